"""Row-band partition of one frame across GPUs -- pure index arithmetic.

The matching window is anchored top-left (SAD.cpp:39), so the band of output rows [r0, r1) needs
input rows [r0, min(r1 + win - 1, H)): a downward-only halo of win-1 rows, no left/right halo.
"""


def band_ranges(rows, n_bands):
    """Split [0, rows) into n_bands equal bands of ceil(rows/n_bands) rows (the last ones may be
    short or empty): equal counts are what an all-gather needs."""
    per = -(-rows // n_bands) if rows > 0 else 0
    out = []
    for b in range(n_bands):
        r0 = min(b * per, rows)
        r1 = min(r0 + per, rows)
        out.append((r0, r1))
    return per, out


def band_input_rows(r0, r1, rows, win):
    """Input rows [r0, r_in_end) needed to produce output rows [r0, r1) of a frame with `rows` rows."""
    return r0, min(r1 + win - 1, rows)
