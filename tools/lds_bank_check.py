#!/usr/bin/env python3
"""Bank-conflict check of the LDS images used by csrc/attention.hip (CPU only, no GPU needed).

Applies the gfx950 banking rules of /opt/skills/guides/MI355X_MICROARCH.md (LDS table): ds_read_b128 is served in four
16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; ds_read_b64 and
ds_read_b64_tr_b16 in the two 32-lane halves; bank of byte address a = (a / 4) % 64.  A group's cost = the largest
number of DISTINCT addresses (identical addresses broadcast) that share one bank.

    python tools/lds_bank_check.py        # prints the worst conflict degree of every read pattern, per head dim
"""
import sys

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
HALVES = [list(range(32)), list(range(32, 64))]


def off(hd, row, ch):
    """byte offset of 16-byte chunk `ch` of row `row` of a [rows][hd] bf16 tile (csrc/attention.hip: lds_off<HD>)"""
    if hd == 128:      # 256-B rows
        return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)))
    if hd == 64:       # 128-B rows, two per bank row
        return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)))
    if hd == 32:       # 64-B rows, four per bank row
        return 64 * row + 16 * (ch ^ ((row >> 2) & 3))
    raise ValueError(hd)


def degree(addrs, groups, width):
    worst = 1
    for g in groups:
        banks = {}
        for lane in g:
            a = addrs[lane]
            for w in range(0, width, 4):
                banks.setdefault(((a + w) // 4) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def row_read(hd, row0, s):
    """ds_read_b128 of the 32x32x16 row operand: lane (r = lane & 31, h = lane >> 5) reads chunk 2s+h of row row0+r"""
    return [off(hd, row0 + (l & 31), 2 * s + (l >> 5)) for l in range(64)]


def tr_read(hd, row0, s, dt, second):
    """ds_read_b64_tr_b16 of the transposed operand in the accumulator-as-operand k order:
    block rows row0 + 16 s + 4 h (+8 for the second read) + q, columns 32 dt + 16 g + 4 p .. +3"""
    out = []
    for l in range(64):
        h, g, i = l >> 5, (l >> 4) & 1, l & 15
        q, p = i >> 2, i & 3
        row = row0 + 16 * s + 4 * h + (8 if second else 0) + q
        c0 = 4 * dt + 2 * g
        out.append(off(hd, row, c0 + (p >> 1)) + 8 * (p & 1))
    return out


def lane_forms_ok(hd):
    """the per-lane XOR forms of csrc/attention.hip (LaneAddr) against off() for every lane and compile-time index"""
    rowb_ = hd * 2
    for lane in range(64):
        h, g, i = lane >> 5, (lane >> 4) & 1, lane & 15
        q, p = i >> 2, i & 3
        rowb = off(hd, lane & 31, h)
        trb = [off(hd, 4 * h + q + 8 * sec, 2 * g + (p >> 1)) + 8 * (p & 1) for sec in (0, 1)]
        for row0 in (0, 32, 64, 96):
            for s in range(hd // 16):
                if (rowb ^ (32 * s)) + rowb_ * row0 != off(hd, row0 + (lane & 31), 2 * s + h):
                    return False
            for s in range(2):
                for dt in range(hd // 32):
                    for sec in (0, 1):
                        want = off(hd, row0 + 16 * s + 4 * h + q + 8 * sec, 4 * dt + 2 * g + (p >> 1)) + 8 * (p & 1)
                        if (trb[sec] ^ (64 * dt)) + rowb_ * (row0 + 16 * s) != want:
                            return False
    return True


def main():
    ok = True
    for hd in (128, 64, 32):
        worst_row = max(degree(row_read(hd, r0, s), B128_GROUPS, 16) for r0 in range(0, 64, 32) for s in range(hd // 16))
        worst_tr = max(degree(tr_read(hd, r0, s, dt, sec), HALVES, 8)
                       for r0 in range(0, 64, 32) for s in range(2) for dt in range(hd // 32) for sec in (0, 1))
        # the image must be a bijection of the tile's bytes
        seen = {off(hd, r, c) for r in range(64) for c in range(hd // 8)}
        bij = len(seen) == 64 * (hd // 8) and max(seen) == 64 * hd * 2 - 16
        forms = lane_forms_ok(hd)
        print(f"hd {hd:3d}: ds_read_b128 rows {worst_row}-way, ds_read_b64_tr_b16 {worst_tr}-way, bijective {bij}, "
              f"lane XOR forms == lds_off: {forms}")
        ok &= bij and forms
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
