"""The tiler on the GPU (leopard_amd/gpu_tiler.py, lmi_resample_u8) against the host tiler (PIL): same u8 tiles, bit for bit.

CPU half: the kernel logic under the emulator build on small images; GPU half: the C3 sample through the C ABI."""
import numpy as np
import pytest
import torch
from PIL import Image

from leopard_amd import tiler
from leopard_amd.gpu_tiler import GpuTiler
from tests.emu_util import emu_ops


@pytest.fixture(scope="module")
def emu_ops_fixture():
    return emu_ops()


def host_tiles(images):
    vit, plan = tiler.tile_sample([Image.fromarray(im) for im in images])
    return tiler.to_u8_tiles(vit), plan


def noise(seed, w, h):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_resample_taps_match_pil_row_by_row():
    """pil_resample_coeffs restates libImaging's tap computation: resizing an impulse image with PIL shows the taps."""
    for n_in, n_out in [(50, 17), (17, 50), (896, 364), (300, 364), (1344, 1092)]:
        bounds, taps = tiler.pil_resample_coeffs(n_in, n_out)
        assert bounds.shape == (n_out, 2) and taps.shape[0] == n_out
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all()
        # taps of one output sample sum to 2^22 up to rounding of each tap
        assert np.abs(taps.sum(1) - (1 << 22)).max() <= taps.shape[1]


@pytest.mark.parametrize("sizes", [[(500, 300)], [(364, 364)], [(900, 420)], [(200, 777)], [(760, 380), (350, 350), (90, 61)]])
def test_emulated_gpu_tiler_is_bit_exact(emu_ops_fixture, sizes):
    images = [noise(10 + i, w, h) for i, (w, h) in enumerate(sizes)]
    got, plan = GpuTiler(emu_ops_fixture, "cpu").tile_sample(images)
    want, plan_h = host_tiles(images)
    assert plan.canvases == plan_h.canvases and tuple(got.shape) == want.shape
    assert np.array_equal(got.numpy(), want)


def test_emulated_resample_rejects_bad_arguments(emu_ops_fixture):
    src = torch.zeros(4, 4, 3, dtype=torch.uint8)
    b, t = (torch.from_numpy(x) for x in tiler.pil_resample_coeffs(4, 2))
    with pytest.raises(RuntimeError):                            # axis must be 0 or 1
        emu_ops_fixture._check(emu_ops_fixture.lib.lmi_resample_u8(src.data_ptr(), src.data_ptr(), 2, 4, 2, 12, 6, b.data_ptr(), t.data_ptr(),
                                                                   t.shape[1], None))


@pytest.mark.gpu
def test_gpu_tiler_c3_sample_bit_exact():
    from leopard_amd.ops import Ops
    from leopard_amd.synth import synth_image_u8
    images = [synth_image_u8(100 + i, 1344, 896) for i in range(6)]
    got, plan = GpuTiler(Ops(), "cuda:0").tile_sample(images)
    want, plan_h = host_tiles(images)
    assert plan.n_vit_inputs == 42 and plan.canvases == plan_h.canvases
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_tiler_mixed_sizes_bit_exact():
    from leopard_amd.ops import Ops
    images = [noise(1, 1920, 1080), noise(2, 640, 1536), noise(3, 364, 364), noise(4, 333, 77), noise(5, 2500, 1700)]
    got, plan = GpuTiler(Ops(), "cuda:0").tile_sample(images)
    want, _ = host_tiles(images)
    assert np.array_equal(got.cpu().numpy(), want)
