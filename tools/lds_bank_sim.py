#!/usr/bin/env python3
"""Bank-conflict check of the transposing fragment reads of gemm_tn.hip (ds_read_b64_tr_b16: two groups of 32 lanes, 8 bytes per
lane, bank = (addr / 4) % 64; MI355X_MICROARCH.md, LDS table).  Prints the worst number of lanes sharing a bank per group for
every (column tile, wave column) of a row layout: 1 = conflict-free."""


def swz(m):
    return 2 * ((m & 3) | (((m >> 3) & 1) << 2))


def swz_hi(m):                      # 384-byte rows: chunks 16..23 permute among themselves
    return 2 * (((m >> 1) & 1) | (((m >> 3) & 1) << 1))


def phys_chunk(c, m, rowb):
    if rowb == 384 and c >= 16:
        return 16 + ((c - 16) ^ swz_hi(m))
    return c ^ swz(m)


def check(rowb, ncols):
    worst = 0
    for c0 in range(0, ncols, 16):                      # 16-column fragment
        for ks in range(2):
            for r in range(2):                          # the two reads of a fragment: rows m and m + 4
                for half in range(2):
                    banks = {}
                    for lane in range(32 * half, 32 * half + 32):
                        g, q, pp = lane >> 4, (lane >> 2) & 3, lane & 3
                        m = 32 * ks + 8 * g + 4 * r + q
                        col = c0 + 4 * pp
                        addr = m * rowb + phys_chunk(col >> 3, m, rowb) * 16 + (col & 7) * 2
                        for b in (addr // 4 % 64, (addr // 4 + 1) % 64):
                            banks[b] = banks.get(b, 0) + 1
                    worst = max(worst, max(banks.values()))
    return worst


if __name__ == "__main__":
    for rowb, ncols in ((512, 256), (256, 128), (384, 192)):
        print(f"row {rowb} B ({ncols} columns): worst lanes per bank = {check(rowb, ncols)}")
    # the swizzle must be an involution inside a row (the DMA applies it on the source side)
    for m in range(64):
        assert sorted(phys_chunk(c, m, 384) for c in range(24)) == list(range(24))
        assert all(phys_chunk(phys_chunk(c, m, 384), m, 384) == c for c in range(24))
    print("384-byte swizzle: permutation of 0..23 and an involution for every row")
