"""The arithmetic the split-product minibatch kernel (csrc/ppo_trunk_bx.hip, csrc/split3.h) rests on, checked on the CPU through the
oracle's restatement (oracle/split3.py): no GPU, no kernel -- the claims of DESIGN.md section 3 "Round 6" (j)."""
import numpy as np

from oracle import split3 as S


def _samples(seed, n=200000):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32)       # 26 decades, both signs
    x[:64] = np.float32(1.0) + np.arange(64, dtype=np.float32) * np.float32(2.0 ** -23)      # neighbours of 1: every low-bit pattern
    x[64:72] = [0.0, -0.0, 1.0, -1.0, 3.0, 0.1, 2.0 ** -100, -(2.0 ** 100)]
    return x


def test_three_bf16_parts_sum_to_the_float32_bit_for_bit():
    x = _samples(1)
    h, m, l = S.split3(x)
    for part in (h, m, l):                                   # each part IS a bf16 number: its low 16 bits are zero
        assert not np.any(part.view(np.uint32) & np.uint32(0xFFFF))
    back = ((h + m).astype(np.float32) + l).astype(np.float32)
    assert np.array_equal(back.view(np.uint32)[x != 0], x.view(np.uint32)[x != 0])
    assert np.all(back[x == 0] == 0)
    ah = np.abs(h.astype(np.float64))
    assert np.all(np.abs(m) <= ah * 2.0 ** -8) and np.all(np.abs(l) <= ah * 2.0 ** -16)


def test_six_part_products_are_the_float32_product_to_two_to_the_minus_23():
    x, y = _samples(2), _samples(3)
    exact = x.astype(np.float64) * y.astype(np.float64)
    six = S.product6(x, y)
    ok = exact != 0
    rel = np.abs(six[ok] - exact[ok]) / np.abs(exact[ok])
    assert rel.max() <= 2.0 ** -23, rel.max()
    # (what that is next to: ONE float32 rounding of the product is up to 2^-24, and a 128-term float32 accumulation carries ~50 of those)
    assert np.median(rel) < 2.0 ** -26


def test_a_128_term_dot_product_from_parts_is_as_close_to_exact_as_a_float32_fma_chain():
    rng = np.random.default_rng(4)
    a = rng.standard_normal((4096, 128)).astype(np.float32)
    b = rng.standard_normal((4096, 128)).astype(np.float32)
    exact = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    # six part products per term, summed in float64: isolates what the DROPPED products cost (the accumulation itself is float32 on both sides)
    parts = S.product6(a, b).sum(1)
    chain = np.zeros(4096, np.float32)
    for k in range(128):                                      # float32 accumulation of exactly-rounded products: what any float32 evaluation pays
        chain = (chain.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
    scale = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(1)
    e_parts, e_chain = np.abs(parts - exact) / scale, np.abs(chain.astype(np.float64) - exact) / scale
    assert e_parts.max() < 2e-8 and np.sqrt(np.mean(e_parts ** 2)) * 10 < np.sqrt(np.mean(e_chain ** 2))
