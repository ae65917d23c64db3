"""Known answers for per-bin coverage, worked out BY HAND from cooltools' documented semantics (cooltools.api.coverage:
every pixel adds its count to BOTH of its bins — a main-diagonal pixel therefore twice; pixels with
|bin2_id - bin1_id| < ignore_diags count as zero, on GLOBAL bin ids, trans pixels included; cov_cis only takes pixels whose
two bins share a chromosome, cov_tot takes all).  cooltools itself is not in the image: these literals are what pins K3
(pup_coverage) and oracle.coverage_numpy to something other than each other.

Matrix (upper triangle; chrA = bins 0..2, chrB = bins 3..4):

          0   1   2 |  3   4
      0   5   2   1 |  6   .
      1       3   4 |  .   .
      2           7 | 11   8
      -------------------------
      3               9  10
      4                   .
"""
import numpy as np

CHROM_OFFSET = np.array([0, 3, 5], np.int64)
PIXELS = [(0, 0, 5), (0, 1, 2), (0, 2, 1), (0, 3, 6), (1, 1, 3), (1, 2, 4), (2, 2, 7), (2, 3, 11), (2, 4, 8), (3, 3, 9), (3, 4, 10)]

# ignore_diags -> (cov_cis, cov_tot), by hand:
#   0: bin0 cis 2*5+2+1 = 13, tot 13+6 = 19;  bin1 2+2*3+4 = 12, 12;  bin2 cis 1+4+2*7 = 19, tot 19+11+8 = 38;
#      bin3 cis 2*9+10 = 28, tot 28+6+11 = 45;  bin4 cis 10, tot 10+8 = 18
#   1: the main diagonal is gone: bin0 cis 2+1 = 3, tot 3+6 = 9;  bin1 2+4 = 6, 6;  bin2 cis 1+4 = 5, tot 5+11+8 = 24;
#      bin3 cis 10, tot 10+6+11 = 27;  bin4 cis 10, tot 18
#   2: only |d| >= 2 is left: (0,2)=1, (0,3)=6, (2,4)=8 — the trans pixel (2,3) at distance 1 is dropped too:
#      bin0 cis 1, tot 7;  bin1 0, 0;  bin2 cis 1, tot 9;  bin3 cis 0, tot 6;  bin4 cis 0, tot 8
#   3: only (0,3)=6 is left: tot bin0 6, bin3 6; cis all zero
ANSWERS = {
    0: ([13, 12, 19, 28, 10], [19, 12, 38, 45, 18]),
    1: ([3, 6, 5, 10, 10], [9, 6, 24, 27, 18]),
    2: ([1, 0, 1, 0, 0], [7, 0, 9, 6, 8]),
    3: ([0, 0, 0, 0, 0], [6, 0, 0, 6, 0]),
}


def table():
    """(bin1_offset, bin2_id, count) of the matrix above, as a .cool stores it."""
    px = sorted(PIXELS)
    rows = np.array([p[0] for p in px])
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=5))]).astype(np.int64)
    return indptr, np.array([p[1] for p in px], np.int32), np.array([p[2] for p in px], np.int32)
