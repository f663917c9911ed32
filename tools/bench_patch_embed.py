"""lmi_patch_embed (fused normalise + im2col + patch conv + bias + pos-emb) against the unfused pair it replaced
(lmi_preprocess_tiles + lmi_gemm) at the C3 shape: 42 ViT inputs of 364 x 364 -> 28392 x 1152.  Run on an MI355X."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402
from leopard_amd.weights import patch_weight_image_order  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ops = Ops()
    n, S, P, N = 42, 364, 14, 1152
    G = S // P
    g = torch.Generator(device=DEV).manual_seed(1)
    u8 = torch.randint(0, 256, (n, S, S, 3), generator=g, device=DEV, dtype=torch.uint8)
    pix = ((u8.float() * (1.0 / 255.0) - 0.5) * 2.0).permute(0, 3, 1, 2).contiguous()
    for dtype in (torch.float16, torch.bfloat16):
        w = (torch.randn(N, 3, P, P, generator=g, device=DEV) * 0.05).to(dtype)
        bias, pos = torch.randn(N, generator=g, device=DEV), torch.randn(G * G, N, generator=g, device=DEV)
        wf = patch_weight_image_order(w, P)
        out = torch.empty(n * G * G, N, device=DEV)
        w2 = torch.zeros(N, 640, dtype=dtype, device=DEV)
        w2[:, :588] = w.reshape(N, -1)
        patches = torch.empty(n * G * G, 640, dtype=dtype, device=DEV)

        def pair():
            ops.preprocess_tiles(u8, patches, S, P)
            ops.gemm(patches, w2, out, bias=bias, addmat=pos, epilogue=_lib.EPI_STORE_F32)
        t_u8 = timed(lambda: ops.patch_embed(u8, wf, bias, pos, out, S, P))
        t_f32 = timed(lambda: ops.patch_embed(pix, wf, bias, pos, out, S, P))
        t_pair = timed(pair)
        t_im = timed(lambda: ops.preprocess_tiles(u8, patches, S, P))
        fl = 2.0 * n * G * G * N * 588
        print(f"{str(dtype):15s} fused from u8 {t_u8:7.1f} us ({fl / t_u8 / 1e6:5.0f} TF/s)   fused from fp32 pixel_values {t_f32:7.1f} us   "
              f"unfused pair {t_pair:7.1f} us (im2col {t_im:.1f} + GEMM {t_pair - t_im:.1f});  output 131 MB -> {131e6 / t_u8 / 1e6:.2f} TB/s written")


if __name__ == "__main__":
    main()
