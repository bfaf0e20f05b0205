"""CPU-side search for an LDS image of the attention tiles ([rows][64] bf16) that is bank-conflict-free for BOTH
access patterns of attention.hip, under the bank rules of MI355X_MICROARCH.md (section LDS):

  * MmaRows: `ds_read_b128`, lane l reads 16 B of row (l & 31), logical 16-byte chunk 2 s + (l >> 5); serviced in the
    four non-contiguous 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63};
  * MmaCols: `ds_read_b64_tr_b16`, serviced in two 32-lane halves; lane l addresses row R + ((l & 15) >> 2) and the
    8 bytes at logical column 2 * (d0 + 16 * ((l >> 4) & 1) + 4 * (l & 3)) of that row.

bank(byte address a) = (a / 4) mod 64 for both; a group conflicts when two lanes touch the same bank with different
addresses.  The image is row * pitch + ((chunk ^ key(row)) * 16) with key a XOR-linear function of the row bits.

The current layout (pitch 144, no swizzle) shows 2-way conflicts in the transpose reads.  Whatever this prints is a
CANDIDATE: the guide warns of further, undocumented conflict classes for the transpose read, so a layout only counts
once SQ_LDS_BANK_CONFLICT confirms it on the GPU."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def conflicts(accesses):
    """accesses: list of (byte address, bytes).  worst number of distinct addresses on one bank"""
    banks = {}
    for a, n in accesses:
        for w in range(a // 4, (a + n) // 4):
            banks.setdefault(w % 64, set()).add(w)
    return max(len(v) for v in banks.values())


def addr(row, chunk, pitch, key):
    return row * pitch + ((chunk ^ key(row)) << 4)


def worst(pitch, key):
    w = 1
    for s in range(4):                      # ds_read_b128 fragment reads, k-step s
        for g in B128_GROUPS:
            w = max(w, conflicts([(addr(32 * 0 + (l & 31), 2 * s + (l >> 5), pitch, key), 16) for l in g]))
    wt = 1
    for R in range(0, 32, 4):               # transpose reads: 4-row block R, column block d0
        for d0 in (0, 32):
            acc = []
            for l in range(32):
                row = R + ((l & 15) >> 2)
                col_b = 2 * (d0 + 16 * ((l >> 4) & 1) + 4 * (l & 3))
                acc.append((addr(row, col_b >> 4, pitch, key) + (col_b & 15), 8))
            wt = max(wt, conflicts(acc))
    return w, wt


def main():
    print('current: pitch 144, no swizzle ->', worst(144, lambda r: 0))
    found = []
    for pitch in (128, 144, 160, 192):
        # key bit j = parity of (row & mask_j), masks over row bits 0..4
        for masks in itertools.product(range(32), repeat=3):
            key = lambda r, m=masks: sum((bin(r & m[j]).count('1') & 1) << j for j in range(3))
            w = worst(pitch, key)
            if w == (1, 1):
                found.append((pitch, masks))
        print(f'pitch {pitch}: {sum(1 for p, _ in found if p == pitch)} conflict-free XOR-linear swizzles')
    for pitch, masks in found[:8]:
        print(f'  pitch {pitch}: key bits = parity(row & {masks[0]:#04x}), parity(row & {masks[1]:#04x}), parity(row & {masks[2]:#04x})')


if __name__ == '__main__':
    main()
