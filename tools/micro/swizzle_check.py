"""LDS bank-conflict check of ds_read_b128 fragment reads (developer tool, CPU only).

ds_read_b128 is serviced in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63};
MI355X_MICROARCH.md, LDS table); a group is conflict-free when its sixteen 16-B accesses fall on sixteen different (address / 16) mod 16.
Rows are 128 B (8 chunks of 16 B), lane (n = lane & 15, g = lane >> 4) reads logical chunk 4 s + g of row R(n, j), stored at chunk ^ f(row).

Checked here: gemm6.hip's A / B slots (consecutive rows, f = (row >> 1) & 7) and gemm6q.hip's B slots (column-permuted rows
R(n, j) = 8 (n >> 2) + 4 j + (n & 3), f = (row & 3) | ((row >> 1) & 4)); the old f on the permuted rows is 2-way conflicted."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def conflict_free(f, rowfn):
    for wc in range(4):
        for j in range(2):
            for s in range(2):
                for grp in GROUPS:
                    keys = set()
                    for lane in grp:
                        g, n = lane >> 4, lane & 15
                        row = wc * 32 + rowfn(n, j)
                        keys.add(((row & 1) * 8 + ((4 * s + g) ^ f(row))) % 16)
                    if len(keys) != 16:
                        return False
    return True


def consecutive(n, j):
    return 16 * j + n


def permuted(n, j):
    return 8 * (n >> 2) + 4 * j + (n & 3)


def f_old(row):
    return (row >> 1) & 7


def f_new(row):
    return (row & 3) | ((row >> 1) & 4)


if __name__ == '__main__':
    print('consecutive rows, f = (row >> 1) & 7                 :', conflict_free(f_old, consecutive))
    print('permuted rows,    f = (row >> 1) & 7                 :', conflict_free(f_old, permuted))
    print('permuted rows,    f = (row & 3) | ((row >> 1) & 4)   :', conflict_free(f_new, permuted))
    print('consecutive rows, f = (row & 3) | ((row >> 1) & 4)   :', conflict_free(f_new, consecutive))
