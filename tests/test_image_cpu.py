"""CPU: pin oracle/resize_oracle.py (Pillow's 8-bit bicubic resample restated) against Pillow itself,
and the host-side size rules against the reference's formulas (data/base_dataset.py:141-168)."""
import numpy as np
import pytest

import resize_oracle as R

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

CASES = [((37, 53), (21, 29)), ((64, 48), (128, 96)), ((50, 50), (50, 31)), ((31, 77), (64, 77)), ((200, 120), (33, 17)),
         ((17, 19), (512, 16)), ((96, 96), (96, 96))]


@pytest.mark.parametrize("src,dst", CASES)
def test_resize_oracle_is_pillow_bit_exact(src, dst):
    rng = np.random.default_rng(src[0] * 1000 + dst[1])
    a = rng.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    a[:4, :4] = 255; a[-4:, -4:] = 0        # saturating ringing at edges
    ref = np.asarray(Image.fromarray(a).resize((dst[1], dst[0]), Image.BICUBIC))
    got = R.resize_u8(a, dst[0], dst[1])
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_size_rules_follow_the_reference_formulas():
    assert R.scale_shortside_size(1024, 768, 512) == (round(1024 * 512 / 768), 512)
    assert R.scale_shortside_size(500, 333, 512) == (round(500 * (512 / 333)), round(333 * (512 / 333)))
    assert R.make_power_2_size(683, 512) == (688, 512)
    assert R.make_power_2_size(520, 520) == (512, 512)      # 520/16 = 32.5 -> Python round() goes to the even 32
    assert R.make_power_2_size(8, 9) == (0, 16)              # the reference formula can collapse tiny images


def test_preprocess_matches_pillow_pipeline():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    t = R.preprocess(a, 64)
    im = Image.fromarray(a)
    w1, h1 = R.scale_shortside_size(130, 90, 64)
    im = im.resize((w1, h1), Image.BICUBIC)
    w2, h2 = R.make_power_2_size(w1, h1)
    if (w2, h2) != (w1, h1):
        im = im.resize((w2, h2), Image.BICUBIC)
    ref = (np.asarray(im).astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    assert t.shape == (3, h2, w2) and np.array_equal(t, ref.transpose(2, 0, 1))


@pytest.mark.parametrize("n_in,n_out", [(53, 29), (48, 96), (683, 688), (1024, 512), (17, 512), (512, 512)])
def test_product_coefficient_tables_equal_the_oracle(n_in, n_out):
    """ppst_amd/imageio.py builds the tables the HIP kernels consume: same integers as the oracle's restatement."""
    from ppst_amd import imageio
    ks, bnd, cf = imageio.resample_tables(n_in, n_out, "cpu")
    oks, obnd, okk = R.coeffs(n_in, n_out)
    assert ks == oks
    assert np.array_equal(bnd.numpy(), np.array(obnd, dtype=np.int32))
    assert np.array_equal(cf.numpy(), np.array(okk, dtype=np.int32))
    assert imageio.scale_shortside_size(500, 333, 512) == R.scale_shortside_size(500, 333, 512)
    assert imageio.make_power_2_size(683, 512) == R.make_power_2_size(683, 512)
