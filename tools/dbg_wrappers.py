import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from quantum_inferno_amd import styx_cwt, styx_stx, engine, synth
x = synth.log_chirp(1 << 20, 1000.0, 0, 1, np.float32)
for mode in ("reference", "native"):
    engine.NUMPY_RESULT_DTYPE = mode
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = styx_cwt.cwt_complex_any_scale_pow2(3, x, 1000.0)[2]
        s = styx_stx.stx_complex_any_scale_pow2(3, x, 1000.0)[2]
        dt = time.perf_counter() - t0
    print(mode, c.dtype, round(dt * 1e3, 1), "ms", torch.cuda.memory.host_memory_stats().get("allocated_bytes.current"))
# staged path
engine.PINNED_RESULT_MAX_BYTES = 1 << 26
engine.NUMPY_RESULT_DTYPE = "reference"
c2 = styx_cwt.cwt_complex_any_scale_pow2(3, x, 1000.0)[2]
print("staged equal:", np.array_equal(c, c2) if c.dtype == c2.dtype else np.allclose(c, c2))
