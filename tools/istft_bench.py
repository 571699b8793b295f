#!/usr/bin/env python3
"""The inverse ShortTimeFFT-convention transform (istft_tukey) at a configs[2]-like shape: fused kernel against the
three-kernel path (QI_TUNE=1 QI_STFT_FUSED=0) -- run once per setting, prints time and a checksum.
tools/istft_bench.py [channels] [segment] [overlap] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantum_inferno_amd.utilities import short_time_fft as stf  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ov = int(sys.argv[3]) if len(sys.argv) > 3 else seg // 2
cdt = torch.complex128 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.complex64
n_slices = (1 << 20) // (seg - ov) + 1
g = torch.Generator(device="cpu").manual_seed(5)
s = torch.randn((ch, seg // 2 + 1, n_slices), generator=g, dtype=torch.float64)
s = torch.complex(s, torch.randn(s.shape, generator=g, dtype=torch.float64)).to(cdt).cuda()
for _ in range(3):
    ts, x = stf.istft_tukey(s, 1000.0, 0.25, seg, ov)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    ts, x = stf.istft_tukey(s, 1000.0, 0.25, seg, ov)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = s.numel() * s.element_size() + x.numel() * x.element_size()
print(f"istft {ch} x {tuple(s.shape[1:])} seg {seg} overlap {ov} {str(cdt)[6:]}: {ms:.4f} ms per call, {nbytes / ms / 1e6:.0f} GB/s of "
      f"required bytes; checksum {float(x.double().abs().sum()):.9e} max {float(x.abs().max()):.6e}")
