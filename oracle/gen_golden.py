"""
Golden-vector generator -- TEST INFRASTRUCTURE, runs ONLY in the build container.

Imports the unmodified reference from /root/reference (PYTHONPATH is set here, the
reference is never copied or shipped), feeds it this build's own seeded synthetic
inputs (oracle.tfr_oracle.synth_chirp) and writes inputs + reference outputs as
small .npz fixtures under tests/golden/.  The GPU box only ever sees the .npz files.

    python oracle/gen_golden.py            # small + medium fixtures (~1 min)
    python oracle/gen_golden.py --large    # adds n = 2^16 and the config-2 size n = 2^20 (~4 min, ~10 GB RSS)
"""
import argparse
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import scipy  # noqa: E402
from quantum_inferno import cwt_atoms, scales_dyadic, styx_cwt, styx_fft, styx_stx, tfr_info  # noqa: E402
from quantum_inferno.utilities import calculations, rescaling  # noqa: E402

from oracle.tfr_oracle import synth_chirp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
VERSIONS = np.array([np.__version__, scipy.__version__, "quantum-inferno 1.1.3"])


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, versions=VERSIONS, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrays)} arrays")


def gen_bands():
    d = {}
    # the reference's own (commented-out) known-answer test, tests/test_scales_dyadic.py:8-21
    d["kat_100hz_8192_n6"] = scales_dyadic.log_frequency_hz_from_fft_points(100.0, 8192, 6, 1.0, scales_dyadic.Slice.G3)
    combos = []
    for fs in (80.0, 800.0, 1000.0, 8000.0, 48000.0):
        for log2n in (8, 10, 13, 16, 20):
            for order in (1, 3, 6, 12, 24):
                n = 2 ** log2n
                f = scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
                if len(f) == 0:
                    continue
                key = f"fs{int(fs)}_n{log2n}_o{order}"
                combos.append(key)
                d[f"f_{key}"] = f
                freq = np.fft.fftfreq(n, 1 / fs)
                d[f"idx_{key}"] = np.array([np.abs(freq - fj).argmin() for fj in f], dtype=np.int64)
                s, w = scales_dyadic.scale_from_frequency_hz(order, f, fs)
                d[f"scale_{key}"] = s
                # cwt_atoms band table (cwt_chirp_from_sig's call chain)
                _, fmin = cwt_atoms.chirp_scales_from_duration(order, n / fs, 0.0, scales_dyadic.Slice.G2)
                out = quiet(
                    cwt_atoms.chirp_frequency_bands,
                    scale_order_input=order,
                    frequency_low_input=fmin,
                    frequency_sample_rate_input=fs,
                    frequency_high_input=fs / 2.0,
                )
                d[f"chirpf_{key}"] = np.flip(out[4])
                d[f"chirpmq_{key}"] = np.array(out[:4], dtype=np.float64)
    d["combos"] = np.array(combos)
    # STFT segment sizing (styx_fft.py:31-41) and rounding helpers
    seg = []
    for fs in (80.0, 800.0, 1000.0, 8000.0, 48000.0):
        for order in (1, 3, 6, 12, 24):
            dur = scales_dyadic.cycles_from_order(order) / (fs * 0.075 / 4)
            seg.append((fs, order, 2 ** calculations.get_num_points(fs, dur, "ceil", "log2")))
    d["stft_seg"] = np.array(seg, dtype=np.float64)
    d["cycles"] = np.array([scales_dyadic.cycles_from_order(o) for o in (0.5, 0.75, 1, 3, 6, 12, 24)])
    d["mqg"] = np.array([cwt_atoms.chirp_mqg_from_n(o) for o in (1, 3, 6, 12, 24)])
    d["log2eps_pm100"] = np.array([rescaling.to_log2_with_epsilon(100.0), rescaling.to_log2_with_epsilon(-100.0)])
    save("bands.npz", **d)


def gen_small():
    """n = 1024: full panels."""
    d = {}
    n = 1024
    for order, fs in ((3, 1000.0), (12, 800.0)):
        key = f"o{order}_fs{int(fs)}"
        sig = synth_chirp(n, fs, dtype=np.float64)
        d[f"sig_{key}"] = sig
        for dt in ("norm", "spect") if order == 3 else ("norm",):
            f, t, cwt = styx_cwt.cwt_complex_any_scale_pow2(order, sig, fs, dictionary_type=dt)
            d[f"cwt_{dt}_{key}"] = cwt
        d[f"f_{key}"] = f
        d[f"t_{key}"] = t
        atoms, t_c, scale, omega, amp = styx_cwt.wavelet_centered_4cwt(order, n, f, fs, "norm")
        d[f"atoms_{key}"] = atoms[[0, len(f) // 2, len(f) - 1]]
        d[f"atom_scale_{key}"] = scale[:, 0]
        d[f"atom_amp_{key}"] = amp[:, 0]
        f2, t2, stx = styx_stx.stx_complex_any_scale_pow2(order, sig, fs)
        assert np.array_equal(f, f2)
        d[f"stx_{key}"] = stx
        c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, order)
        d[f"chirp_cwt_{key}"] = c
        d[f"chirp_bits_{key}"] = bits
        d[f"chirp_f_{key}"] = fc
        if order == 3:
            c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, order, dictionary_type="spect")
            d[f"chirp_cwt_spect_{key}"] = c
            c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, order, cwt_type="conv")
            d[f"chirp_cwt_conv_{key}"] = c
    # tfr_info on the order-3 CWT power panel (tutorial convention power = 2|z|^2, s04_tone_tfr.py:92)
    p = 2 * np.abs(d["cwt_norm_o3_fs1000"]) ** 2
    d["info_power"] = p
    a, b, c = tfr_info.power_dynamics_scaled_bits(p)
    d["info_bits"], d["info_bits_time"], d["info_bits_freq"] = a, b, c
    for nm, obj in (
        ("tot", tfr_info.shannon_stft_from_tfr_power(p)),
        ("time", tfr_info.ShannonStftPerTime(p)),
        ("freq", tfr_info.ShannonStftPerFreq(p)),
    ):
        d[f"sh_{nm}_info"] = obj.info
        d[f"sh_{nm}_bits"] = obj.shannon_bits
        d[f"sh_{nm}_ref"] = np.array(obj.ref_bits)
        d[f"sh_{nm}_isnr"] = obj.isnr
        d[f"sh_{nm}_esnr"] = obj.esnr
    tdr, fftc = tfr_info.shannon_tdr_fft(d["sig_o3_fs1000"])
    for nm, obj in (("tdr", tdr), ("fft", fftc)):
        d[f"sh1_{nm}_info"], d[f"sh1_{nm}_entropy"] = obj.info, obj.entropy
        d[f"sh1_{nm}_isnr"], d[f"sh1_{nm}_esnr"] = obj.isnr, obj.esnr
        d[f"sh1_{nm}_ref"] = np.array(obj.ref_entropy)
    save("small_n1024.npz", **d)


def gen_stx_general():
    """styx_stx.tfr_stx_fft on its working subset (n_fft_in = len(sig), a power of two)."""
    d = {}
    n, fs = 1024, 1000.0
    sig = synth_chirp(n, fs, dtype=np.float64)
    d["sig"] = sig
    cases = {
        "lin": dict(frequency_min=20.0, frequency_max=400.0, frequency_step=20.0),
        "geo": dict(scale_order_input=3.0, frequency_min=10.0, frequency_max=450.0, is_geometric=True),
        "inferno": dict(scale_order_input=3.0, frequency_min=8.0, frequency_max=400.0, is_geometric=True, is_inferno=True),
        "qpr": dict(frequency_min=25.0, frequency_max=300.0, frequency_step=25.0, factor_q=0.5, power_p=1.0, power_r=0.75),
    }
    for name, kw in cases.items():
        tfr, psd, f, f_fft, win = quiet(styx_stx.tfr_stx_fft, sig, 1 / fs, n_fft_in=n, **kw)
        d[f"{name}_tfr"], d[f"{name}_f"], d[f"{name}_ffft"] = tfr, f, f_fft
        d[f"{name}_psd_row0"], d[f"{name}_win_rows"] = psd[0], win[[0, len(f) - 1]]
    save("stx_general_n1024.npz", **d)


def gen_stft():
    d = {}
    for log2n, fs in ((13, 1000.0), (13, 800.0), (16, 1000.0)):
        n = 2 ** log2n
        for dtype in (np.float64, np.float32):
            sig = synth_chirp(n, fs, dtype=dtype)
            tag = f"n{log2n}_fs{int(fs)}_{np.dtype(dtype).name}"
            d[f"sig_{tag}"] = sig
            for order in (3, 12):
                z, bits, t, f = styx_fft.stft_from_sig(sig, fs, order)
                cols = slice(None) if (log2n == 13 or (order == 3 and dtype == np.float64)) else slice(0, None, 8)
                d[f"z_{tag}_o{order}"] = z[:, cols]
                d[f"bits_{tag}_o{order}"] = bits[:, cols].astype(np.float32) if log2n == 16 else bits[:, cols]
                d[f"t_{tag}_o{order}"] = t
                d[f"f_{tag}_o{order}"] = f
                d[f"shape_{tag}_o{order}"] = np.array(z.shape)
    # stft_complex_pow2 with its own default alpha = 0.25 Tukey, 2-D input (axis -1)
    for dtype in (np.float64, np.float32):
        sigw = synth_chirp(8192, 1000.0, dtype=dtype)
        fw, pw = styx_fft.welch_power_pow2(sigw, 1000.0, 512)
        d[f"welch_f_{np.dtype(dtype).name}"], d[f"welch_p_{np.dtype(dtype).name}"] = fw, pw
        fw, pw = styx_fft.welch_power_pow2(sigw, 1000.0, 300, nfft_points=512, overlap_points=100, alpha=0.5)
        d[f"welch2_p_{np.dtype(dtype).name}"] = pw
    sig2 = np.stack([synth_chirp(4096, 1000.0, c, 3, np.float64) for c in range(3)])
    f, t, z = styx_fft.stft_complex_pow2(sig2, 1000.0, 256)
    d["sig_2d"], d["z_2d_alpha025"], d["t_2d"], d["f_2d"] = sig2, z, t, f
    save("stft.npz", **d)


def gen_stft_large():
    """The STFT of configs[2]: 2^20 samples @ 1000 Hz, order 12 (2048-sample segments, 1025 x 1025 bins): sampled rows and
    columns of the reference's panel and bits for a float32 and a float64 record, the axes and the panel maximum."""
    n, fs, order = 1 << 20, 1000.0, 12
    d = {}
    for dtype in (np.float32, np.float64):
        tag = np.dtype(dtype).name
        sig = synth_chirp(n, fs, dtype=dtype)
        z, bits, t, f = styx_fft.stft_from_sig(sig, fs, order)
        rows = np.unique(np.concatenate([np.arange(0, z.shape[0], 64), [1, z.shape[0] - 2, z.shape[0] - 1]]))
        cols = np.unique(np.concatenate([np.arange(0, z.shape[1], 64), [1, z.shape[1] - 2, z.shape[1] - 1]]))
        d[f"rows_{tag}"], d[f"cols_{tag}"] = rows, cols
        d[f"z_rows_{tag}"], d[f"z_cols_{tag}"] = z[rows], z[:, cols]
        d[f"bits_rows_{tag}"], d[f"bits_cols_{tag}"] = bits[rows], bits[:, cols]
        d[f"zmax_{tag}"] = np.abs(z).max()
        d[f"shape_{tag}"] = np.array(z.shape)
        d[f"t_{tag}"], d[f"f_{tag}"] = t, f
        d[f"sig_samples_{tag}"] = sig[:: n // 4096]
    save("stft_n1048576_o12.npz", **d)


def time_samples(n, dense=False):
    """Time samples kept of a long panel row: a regular comb, the middle of the record and -- `dense` (the benchmark
    length, every band kept) -- both record ends (zero padding / wrap-around act there) and a few samples around
    multiples of the overlap-save block strides of the block engine (3584, 3072, 2048 outputs per block)."""
    parts = [np.arange(0, n, max(1, n // 512)), np.arange(n // 2 - 256, n // 2 + 256)]
    if dense:
        parts += [np.arange(0, 96), np.arange(n - 96, n)]
        for stride, k in ((3584, 100), (3072, 171), (2048, 300), (3584, 292), (2048, 511)):
            parts.append(np.arange(stride * k - 6, stride * k + 6))
    t = np.unique(np.concatenate(parts))
    return t[(t >= 0) & (t < n)]


def panel_digest(panel, rows, dense=False):
    p = np.abs(panel) ** 2
    sh = tfr_info.shannon_stft_from_tfr_power(p)
    n = panel.shape[1]
    tsel = time_samples(n, dense)
    return {
        "rows": panel[rows][:, tsel] if n > 8192 else panel[rows],
        "tsel": tsel,
        "psum_band": p.sum(axis=1),
        "psum_time": p.sum(axis=0)[tsel] if n > 8192 else p.sum(axis=0),
        "pmax": np.array(p.max()),
        "ptot": np.array(p.sum()),
        "entropy_bits": np.array(np.sum(sh.shannon_bits)),
    }


def gen_short_time_fft():
    """utilities/short_time_fft.py: stft_tukey / spectrogram_tukey / istft_tukey and the complex STFT of the same
    ShortTimeFFT object (the input of the inverse)."""
    from quantum_inferno.utilities import short_time_fft as stf

    d = {}
    rng = np.random.default_rng(7)
    sig = (synth_chirp(4096, 800.0, dtype=np.float64) + 0.1 * rng.standard_normal(4096))
    d["sig"] = sig
    cases = [("a", 800.0, 0.25, 256, 128, "magnitude", "zeros"), ("b", 800.0, 0.5, 200, 150, "psd", "even"),
             ("c", 800.0, 1.0, 128, 96, "magnitude", "odd"), ("d", 800.0, 0.25, 256, 192, "magnitude", "edge")]
    d["cases"] = np.array([f"{c[0]},{c[1]},{c[2]},{c[3]},{c[4]},{c[5]},{c[6]}" for c in cases])
    for tag, fs, alpha, seg, ov, scaling, padding in cases:
        f, t, mag = quiet(stf.stft_tukey, sig, fs, alpha, seg, ov, scaling, padding)
        f2, t2, sxx = quiet(stf.spectrogram_tukey, sig, fs, alpha, seg, ov, scaling, padding)
        obj = quiet(stf.get_stft_object_tukey, fs, alpha, seg, ov, scaling)
        S = obj.stft(sig)
        ts, x = quiet(stf.istft_tukey, S, fs, alpha, seg, ov, scaling)
        d[f"f_{tag}"], d[f"t_{tag}"], d[f"mag_{tag}"], d[f"sxx_{tag}"] = f, t, mag, sxx
        d[f"S_{tag}"], d[f"ts_{tag}"], d[f"x_{tag}"] = S, ts, x
        d[f"geom_{tag}"] = np.array([obj.p_min, obj.p_max(len(sig)), obj.hop, obj.mfft, obj.m_num_mid])
    save("short_time_fft.npz", **d)


def gen_shannon1d():
    """1-D Shannon family (tfr_info.py:97-200) on chirp + noise records (no empty spectrum bins, so that the unwrapped
    phase is well defined), float64 and float32."""
    d = {}
    rng = np.random.default_rng(20250213)
    for tag, dt, n in (("f64", np.float64, 4096), ("f32", np.float32, 4096), ("f64_odd", np.float64, 1000)):
        sig = (synth_chirp(n, 1000.0, dtype=np.float64) + 0.5 * rng.standard_normal(n)).astype(dt)
        tdr, fft = tfr_info.shannon_tdr_fft(sig)
        d[f"sig_{tag}"] = sig
        for name, obj in (("tdr", tdr), ("fft", fft)):
            d[f"{name}_sig_{tag}"] = obj.sig
            d[f"{name}_marginal_{tag}"] = obj.marginal
            d[f"{name}_info_{tag}"] = obj.info
            d[f"{name}_entropy_{tag}"] = obj.entropy
            d[f"{name}_ref_entropy_{tag}"] = np.array(obj.ref_entropy)
            d[f"{name}_isnr_{tag}"] = obj.isnr
            d[f"{name}_esnr_{tag}"] = obj.esnr
        d[f"fft_angle_{tag}"] = fft.angle_rads
        d[f"fft_frequency_{tag}"] = fft.frequency
    save("shannon1d.npz", **d)


def gen_corners():
    """Three corners of the path that had no reference vector: the Gaussian-window STFT (styx_fft.py:190-227), chirped
    atoms index_shift = +-1 (cwt_atoms.py:22,42-44,202-211) and the unit-amplitude dictionary (styx_cwt.py:139)."""
    d = {}
    n, fs = 2048, 1000.0
    for dtype in (np.float64, np.float32):
        sig = synth_chirp(n, fs, dtype=dtype)
        tag = np.dtype(dtype).name
        d[f"sig_{tag}"] = sig
        f, t, z = styx_fft.gtx_complex_pow2(sig, fs, 256)
        d[f"gtx_f_{tag}"], d[f"gtx_t_{tag}"], d[f"gtx_z_{tag}"] = f, t, z
        f, t, z = styx_fft.gtx_complex_pow2(sig, fs, 200, gaussian_sigma=30, overlap_points=150, nfft_points=512)
        d[f"gtx2_f_{tag}"], d[f"gtx2_t_{tag}"], d[f"gtx2_z_{tag}"] = f, t, z
    sig = d["sig_float64"]
    for shift, tag in ((1.0, "p1"), (-1.0, "m1")):
        c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, 3, index_shift=shift)
        d[f"shift_cwt_{tag}"], d[f"shift_f_{tag}"] = c, fc
        if shift > 0:
            d[f"shift_bits_{tag}"] = bits
        d[f"shift_mqg_{tag}"] = np.array(cwt_atoms.chirp_mqg_from_n(3, shift))
    c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, 6, index_shift=1.0, dictionary_type="spect")
    d["shift_cwt_o6_spect_p1"] = c[::3]
    for order in (3,):
        f, t, cwt = styx_cwt.cwt_complex_any_scale_pow2(order, sig, fs, dictionary_type="unit")
        d[f"unit_cwt_o{order}"], d[f"unit_f_o{order}"] = cwt, f
    atoms, t_c, scale, omega, amp = styx_cwt.wavelet_centered_4cwt(3, n, f[:4], fs, "unit")
    d["unit_atoms"], d["unit_amp"] = atoms, amp[:, 0]
    save("corners_n2048.npz", **d)


def gen_sized(log2n, orders, fs, name, all_rows=False, dtype=np.float32, channel=(0, 1), transforms=("cwt", "stx", "chirp")):
    """all_rows: keep EVERY band (at the sampled times) instead of three, so that each sub-engine of the native path
    (zoom levels, block reach groups, narrow / wide filter spectra, split bands) is pinned by the reference.
    dtype: of the record handed to the reference (float64: the reference then works in double throughout);
    channel = (c, of): record c of a batch of `of` (the phase offset and noise seed of that channel);
    transforms: which of the three panels to capture."""
    d = {}
    n = 2 ** log2n
    sig = synth_chirp(n, fs, channel[0], channel[1], dtype=dtype)
    if n <= 65536:
        d["sig"] = sig
    else:  # regenerated by the tests with the same seeded generator; pinned by samples
        d["sig_samples"] = sig[:: n // 4096]
    for order in orders:
        f = scales_dyadic.log_frequency_hz_from_fft_points(fs, n, order)
        rows = np.arange(len(f)) if all_rows else np.array([0, len(f) // 2, len(f) - 1])
        d[f"f_o{order}"], d[f"rows_o{order}"] = f, rows
        if "cwt" in transforms:
            f1, t, cwt = styx_cwt.cwt_complex_any_scale_pow2(order, sig, fs)
            assert np.array_equal(f1, f)
            for k, v in panel_digest(cwt, rows, all_rows).items():
                d[f"cwt_{k}_o{order}"] = v
            del cwt
            print(f"  n=2^{log2n} order {order} cwt done", flush=True)
        if "stx" in transforms:
            f2, t2, stx = styx_stx.stx_complex_any_scale_pow2(order, sig, fs)
            assert np.array_equal(f2, f)
            for k, v in panel_digest(stx, rows, all_rows).items():
                d[f"stx_{k}_o{order}"] = v
            del stx
        if "chirp" not in transforms:
            print(f"  n=2^{log2n} order {order} done", flush=True)
            continue
        c, bits, tc, fc = quiet(cwt_atoms.cwt_chirp_from_sig, sig, fs, order)
        rows_c = np.arange(len(fc)) if all_rows else np.array([0, len(fc) // 2, len(fc) - 1])
        d[f"chirp_f_o{order}"], d[f"chirp_rowsel_o{order}"] = fc, rows_c
        for k, v in panel_digest(c, rows_c, all_rows).items():
            d[f"chirp_{k}_o{order}"] = v
        del c, bits
        print(f"  n=2^{log2n} order {order} done", flush=True)
    save(name, **d)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    todo = a.only.split(",") if a.only else ["bands", "small", "stft", "stxgen", "shannon1d", "stfft", "corners", "medium"] + (["large"] if a.large else [])
    if "bands" in todo:
        gen_bands()
    if "small" in todo:
        gen_small()
    if "stft" in todo:
        gen_stft()
    if "stftlarge" in todo:
        gen_stft_large()
    if "stxgen" in todo:
        gen_stx_general()
    if "shannon1d" in todo:
        gen_shannon1d()
    if "stfft" in todo:
        gen_short_time_fft()
    if "corners" in todo:
        gen_corners()
    if "medium" in todo:
        gen_sized(13, (3, 12), 1000.0, "medium_n8192.npz")
    if "large" in todo:
        gen_sized(16, (3, 12), 800.0, "large_n65536.npz")
    if "large" in todo or "large20" in todo:
        gen_sized(20, (3,), 1000.0, "large_n1048576.npz", all_rows=True)
    if "large20f64" in todo:  # the same record in float64: the reference works in double throughout (~10 GB RSS)
        gen_sized(20, (3,), 1000.0, "large_n1048576_f64.npz", all_rows=True, dtype=np.float64, transforms=("cwt", "stx"))
    if "cfg3ch63" in todo:  # channel 63 of the BASELINE configs[2] batch, Stockwell only (~3 GB RSS)
        gen_sized(20, (12,), 1000.0, "large_n1048576_o12_ch63_stx.npz", all_rows=True, channel=(63, 64), transforms=("stx",))
    if "o6n19" in todo:  # an order-6 table at 2^19 samples: other zoom classes / reach groups than orders 3 and 12 at 2^20
        gen_sized(19, (6,), 1000.0, "large_n524288_o6.npz", all_rows=True, transforms=("cwt", "stx"))
    if "large12f64" in todo:  # the order-12 table on a FLOAT64 record (the shape of BASELINE configs[4] per chunk and of the
        # bench's f64 leg): every band of both panels from the reference working in double throughout; ~30 GB RSS, ~8 min
        gen_sized(20, (12,), 1000.0, "large_n1048576_o12_f64.npz", all_rows=True, dtype=np.float64, transforms=("cwt", "stx"))
    if "large12" in todo:  # BASELINE configs[2] per channel: order 12 (167 bands) at 2^20 samples; ~30 GB RSS, ~6 min
        gen_sized(20, (12,), 1000.0, "large_n1048576_o12.npz", all_rows=True)
