"""The reference's radix sort, emulated invocation by invocation from the shader text at subgroup size 32
(oracle/radix_glsl.py), against the contract everything else in this repository relies on: after the four passes the
pairs are ascending by the full 32-bit key and equal keys keep their emission order — i.e. a stable sort.  CPU only."""
import numpy as np
import pytest

import oracle
from oracle import radix_glsl as rg
from conftest import make_case, oracle_frame


def _check(keys, values):
    sk, sv = rg.sort_pairs(keys, values)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(sk, keys[order])
    np.testing.assert_array_equal(sv, values[order])
    ok, ov = oracle.sort_pairs(keys, values)       # the C oracle's sort
    np.testing.assert_array_equal(sk, ok)
    np.testing.assert_array_equal(sv, ov)


@pytest.mark.parametrize("d", [1, 2, 31, 33, 511, 4095, 4096, 4097, 10000, 3 * 4096])
def test_literal_radix_sort_is_a_stable_sort(d):
    rng = np.random.default_rng(400 + d)
    values = np.arange(d, dtype=np.uint32)
    _check(rng.integers(0, 2 ** 32, d, dtype=np.uint64).astype(np.uint32), values)
    # few distinct keys: long runs of ties whose order is the point
    _check((rng.integers(0, 7, d).astype(np.uint32) << 16) | rng.integers(0, 3, d).astype(np.uint32), values)
    # every digit of every pass equal but one
    _check(np.full(d, 0x00AB00CD, np.uint32) | (rng.integers(0, 2, d).astype(np.uint32) << 24), values)
    # the padding key itself is a legal key (tile 65535, depth 65535)
    _check(np.where(rng.random(d) < 0.3, 0xFFFFFFFF, rng.integers(0, 2 ** 32, d, dtype=np.uint64)).astype(np.uint32), values)


def test_literal_radix_sort_on_a_frame_of_the_pipeline():
    """The pairs the projection emits for a scene (ascending splat id, gsplat_projection.glsl:219-226) through the
    literal sort: the oracle's sorted arrays — the ones every GPU parity test compares the HIP path with."""
    case = make_case(3000, 320, 192, seed=305, sh_degree=0, scale_n=300)
    ref = oracle.render_frame(case["records"], oracle_frame(case), capacity=400 * case["records"].shape[0])
    assert ref["stats"]["overflow"] == 0 and ref["D"] > 10000
    sk, sv = rg.sort_pairs(ref["keys_unsorted"], ref["values_unsorted"])
    np.testing.assert_array_equal(sk, ref["keys"])
    np.testing.assert_array_equal(sv, ref["values"])
    assert (np.diff(ref["keys"].astype(np.int64)) == 0).mean() > 0.01   # there are ties to keep in order
