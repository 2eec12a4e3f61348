"""TEST INFRASTRUCTURE (oracle/): NumPy restatement of csrc/split3.h -- the exact 3-way bf16 split of a float32 and the six-term
product csrc/ppo_trunk_bx.hip issues on the matrix cores.  Nothing under xuance_amd/ imports this; tests/test_oracle_split3.py pins
the arithmetic claims DESIGN.md section 3 "Round 6" (j) makes (they need no GPU): x == h + m + l bit for bit; |m| <= 2^-8 |h|,
|l| <= 2^-16 |h|; the three dropped part products are <= 2^-23 |x y| per scalar product.

There is no reference file:line for this: the reference multiplies float32 numbers with torch's sgemm (ppo_learner.py:46-62 through
nn.Linear); what is restated here is how the engine forms the SAME float32 products from bf16 parts."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bfloat16 (ties to even), returned as float32 (v_cvt_pk_bf16_f32's rounding for finite, normal values)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    """h, m, l (float32 arrays holding bf16 values) with x == (h + m) + l exactly."""
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    l = bf16_rne((r - m).astype(np.float32))
    return h, m, l


def product6(x, y):
    """The six part products the kernel accumulates, in float64 (each bf16 x bf16 product is exact in float32; the matrix core adds
    them to a float32 accumulator -- the accumulation order is the hardware's, the SET of terms is what is restated)."""
    xh, xm, xl = (a.astype(np.float64) for a in split3(x))
    yh, ym, yl = (a.astype(np.float64) for a in split3(y))
    return xl * yh + xh * yl + xm * ym + xm * yh + xh * ym + xh * yh
