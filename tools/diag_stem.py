"""Time attribution for the stem kernels of the parity mode (diagnostic library only; outputs of the switched-off runs are wrong by design).

  bash simclr_amd/csrc/build.sh diag
  SIMCLR_HIP_LIB=simclr_amd/libsimclr_hip_diag.so python tools/diag_stem.py

  stem_conv_fwd<float, ., 14, 13>:  SIMCLR_DIAG 1 = no MFMA   2 = no activation loads in the tile loop   4 = no stores
  conv_wgrad_dma<float, 256, 64, ...> (stem):  1 = no LDS reads + MFMA   2 = no global loads
  stem backward (max-pool backward + BatchNorm backward): unfused against the fused pair
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from simclr_amd import ops  # noqa: E402
from tools.diag_conv import timeit  # noqa: E402


def main():
    V, H = int(os.environ.get('V', '1024')), 224
    dev = 'cuda'
    ops.set_f32_matmul('f16x3_3')
    geo = ops.stem_geometry(H, H, 7, 7, 2)
    img = torch.rand(V // 2, H, H, 6, device=dev)
    xp = ops.pack_views(img, 2, geo, torch.float32)
    w = torch.randn(7, 7, 3, 64, device=dev) * 0.08
    w_s = ops.prep_weights(w, 2, torch.float32, geo['KHP'], geo['KWP'])
    dy = torch.randn(V, geo['OH'], geo['OW'], 64, device=dev) * 1e-3
    M = V * geo['OH'] * geo['OW']

    def fwd():
        stats = ops.stem_stats(M, 64, dev)
        return ops.stem_conv_fwd(xp, w_s, geo, 2, stats=stats)

    def wgrad():
        return ops.stem_conv_wgrad(xp, dy, geo, 7, 7, 2)

    for name, fn, modes in (('stem_conv_fwd', fwd, [0, 1, 2, 4, 6, 3, 7]), ('stem wgrad', wgrad, [0, 1, 2, 3])):
        out = []
        for m in modes:
            os.environ['SIMCLR_DIAG'] = str(m)
            out.append(timeit(fn))
        os.environ['SIMCLR_DIAG'] = '0'
        print('%-14s ' % name + '  '.join('diag %d: %7.1f us' % (m, t) for m, t in zip(modes, out)), flush=True)


if __name__ == '__main__':
    main()
