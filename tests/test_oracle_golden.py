"""Pins the CPU oracle (oracle/tfr_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, written by oracle/gen_golden.py in the build container) and to the
reference's one known-answer vector for this path.  CPU only."""
import numpy as np
import pytest

from oracle import tfr_oracle as orc
from conftest import relmax


def test_reference_known_answer(golden):
    # reference tests/test_scales_dyadic.py:8-21 (commented out upstream): 100 Hz, 8192 points, order 6
    f = orc.band_table(100.0, 8192, 6, 1.0, orc.G3)
    assert len(f) == 48
    assert f[0] == 0.1778279410038923
    assert f[-1] == 39.810717055349706
    assert np.array_equal(f, golden("bands.npz")["kat_100hz_8192_n6"])


def test_band_tables_bit_exact(golden):
    g = golden("bands.npz")
    for key in g["combos"]:
        fs_s, n_s, o_s = str(key).split("_")
        fs, n, order = float(fs_s[2:]), 2 ** int(n_s[1:]), int(o_s[1:])
        f = orc.band_table(fs, n, order)
        assert np.array_equal(f, g[f"f_{key}"]), key
        assert np.array_equal(orc.stx_indices(f, n, fs), g[f"idx_{key}"]), key
        assert np.array_equal(orc.scale_omega(order, f, fs)[0], g[f"scale_{key}"]), key
        order_n, f_chirp = orc.chirp_band_table(order, n, fs)
        assert np.array_equal(np.flip(f_chirp), g[f"chirpf_{key}"]), key
        mqg = orc.chirp_mqg_from_n(order_n)
        assert np.array_equal(np.array([order_n, mqg[0], mqg[1], mqg[2]]), g[f"chirpmq_{key}"]), key


def test_scalars_bit_exact(golden):
    g = golden("bands.npz")
    for fs, order, seg in g["stft_seg"]:
        assert orc.stft_segment_points(fs, order) == int(seg)
    assert np.array_equal(np.array([orc.cycles_from_order(o) for o in (0.5, 0.75, 1, 3, 6, 12, 24)]), g["cycles"])
    assert np.array_equal(np.array([orc.chirp_mqg_from_n(o) for o in (1, 3, 6, 12, 24)]), g["mqg"])
    assert np.array_equal(orc.to_log2_with_epsilon(np.array([100.0, -100.0])), g["log2eps_pm100"])
    # reference tests/utilities/test_calculations.py:86-100, test_rescaling.py:7-21
    assert orc.get_num_points(10, 10, "round", "log2") == 7
    assert orc.round_value(1.5, "round") == 2 and orc.round_value(1.5, "floor") == 1
    assert abs(orc.to_log2_with_epsilon(100.0) - 6.64) < 0.01


@pytest.mark.parametrize("key,order,fs", [("o3_fs1000", 3, 1000.0), ("o12_fs800", 12, 800.0)])
def test_small_panels(golden, key, order, fs):
    g = golden("small_n1024.npz")
    sig = g[f"sig_{key}"]
    f, t, cwt = orc.cwt_fft(order, sig, fs, "norm")
    assert np.array_equal(f, g[f"f_{key}"]) and np.array_equal(t, g[f"t_{key}"])
    assert relmax(cwt, g[f"cwt_norm_{key}"]) < 1e-14
    sel = [0, len(f) // 2, len(f) - 1]
    atoms = np.stack([orc.gabor_atom_row(order, len(sig), f[j], fs) for j in sel])
    assert relmax(atoms, g[f"atoms_{key}"]) < 1e-15
    s, _ = orc.scale_omega(order, f, fs)
    assert np.array_equal(s, g[f"atom_scale_{key}"])
    assert np.array_equal(orc.wavelet_amplitude(s)[0], g[f"atom_amp_{key}"])
    f2, t2, stx = orc.stx_fft(order, sig, fs)
    assert relmax(stx, g[f"stx_{key}"]) < 1e-14
    c, bits, tc, fc = orc.cwt_chirp_fft(sig, fs, order)
    assert np.array_equal(fc, g[f"chirp_f_{key}"])
    assert relmax(c, g[f"chirp_cwt_{key}"]) < 1e-11
    big = np.abs(g[f"chirp_cwt_{key}"]) > 1e-9
    assert np.max(np.abs(bits - g[f"chirp_bits_{key}"])[big]) < 1e-6
    if order == 3:
        assert relmax(orc.cwt_fft(order, sig, fs, "spect")[2], g[f"cwt_spect_{key}"]) < 1e-14
        assert relmax(orc.cwt_chirp_fft(sig, fs, order, dict_type="spect")[0], g[f"chirp_cwt_spect_{key}"]) < 1e-11


def test_cwt_atoms_conv_backend_matches_away_from_edges(golden):
    # the reference's own cross-check of its circular fft back-end (cwt_atoms.py:423-435)
    g = golden("small_n1024.npz")
    a, b = g["chirp_cwt_o3_fs1000"], g["chirp_cwt_conv_o3_fs1000"]
    assert relmax(a[-12:, 256:768], b[-12:, 256:768]) < 1e-6


def test_tfr_info(golden):
    g = golden("small_n1024.npz")
    p = g["info_power"]
    a, b, c = orc.power_dynamics_scaled_bits(p)
    assert np.array_equal(a, g["info_bits"]) and np.array_equal(b, g["info_bits_time"]) and np.array_equal(c, g["info_bits_freq"])
    for nm, obj in (("tot", orc.shannon_from_power(p)), ("time", orc.shannon_per_time(p)), ("freq", orc.shannon_per_freq(p))):
        assert np.array_equal(obj.info, g[f"sh_{nm}_info"]), nm
        assert np.array_equal(obj.shannon_bits, g[f"sh_{nm}_bits"]), nm
        assert obj.ref_bits == float(g[f"sh_{nm}_ref"])
        assert np.array_equal(obj.isnr, g[f"sh_{nm}_isnr"]) and np.array_equal(obj.esnr, g[f"sh_{nm}_esnr"])
    sig = g["sig_o3_fs1000"]
    x = sig / np.sqrt(np.sum(sig ** 2))
    info, ent, ref, isnr, esnr = orc.shannon_1d(x ** 2)
    assert np.array_equal(info, g["sh1_tdr_info"]) and np.array_equal(ent, g["sh1_tdr_entropy"])
    assert np.array_equal(isnr, g["sh1_tdr_isnr"]) and np.array_equal(esnr, g["sh1_tdr_esnr"])
    sq = np.abs(np.fft.rfft(sig)) ** 2
    info, ent, ref, isnr, esnr = orc.shannon_1d(sq / np.sum(sq))
    assert np.allclose(info, g["sh1_fft_info"], rtol=1e-12, atol=0) and ref == float(g["sh1_fft_ref"])


@pytest.mark.parametrize("tag", ["n13_fs1000", "n13_fs800", "n16_fs1000"])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("order", [3, 12])
def test_stft(golden, tag, dtype, order):
    g = golden("stft.npz")
    fs = float(tag.split("fs")[1])
    sig = g[f"sig_{tag}_{dtype}"]
    z, bits, t, f = orc.stft_from_sig(sig, fs, order)
    key = f"{tag}_{dtype}_o{order}"
    assert np.array_equal(np.array(z.shape), g[f"shape_{key}"])
    assert np.array_equal(t, g[f"t_{key}"]) and np.array_equal(f, g[f"f_{key}"])
    gz = g[f"z_{key}"]
    step = 1 if gz.shape[1] == z.shape[1] else 8
    assert z.dtype == gz.dtype
    assert np.array_equal(z[:, ::step], gz)
    assert np.allclose(bits[:, ::step], g[f"bits_{key}"], rtol=0, atol=1e-5 if tag.startswith("n16") else 0)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_stft_benchmark_shape(golden, dtype):
    """The STFT of BASELINE configs[2] (2^20 samples, order 12: 1025 x 1025 bins) against sampled rows and columns of the
    reference's panel."""
    g = golden("stft_n1048576_o12.npz")
    n, fs, order = 1 << 20, 1000.0, 12
    sig = orc.synth_chirp(n, fs, dtype=np.dtype(dtype).type)
    assert np.array_equal(sig[:: n // 4096], g[f"sig_samples_{dtype}"])
    z, bits, t, f = orc.stft_from_sig(sig, fs, order)
    assert np.array_equal(np.array(z.shape), g[f"shape_{dtype}"])
    assert np.array_equal(t, g[f"t_{dtype}"]) and np.array_equal(f, g[f"f_{dtype}"])
    rows, cols = g[f"rows_{dtype}"], g[f"cols_{dtype}"]
    assert np.array_equal(z[rows], g[f"z_rows_{dtype}"]) and np.array_equal(z[:, cols], g[f"z_cols_{dtype}"])
    assert np.allclose(bits[rows], g[f"bits_rows_{dtype}"], rtol=0, atol=1e-5)


def test_stft_2d_tukey_quarter(golden):
    g = golden("stft.npz")
    f, t, z = orc.stft_complex_pow2(g["sig_2d"], 1000.0, 256)
    assert np.array_equal(f, g["f_2d"]) and np.array_equal(t, g["t_2d"])
    assert np.array_equal(z, g["z_2d_alpha025"])


def test_stft_short_signal_raises():
    with pytest.raises(ValueError):
        orc.stft_from_sig(np.zeros(1024), 1000.0, 12)


def test_medium_digests(golden):
    g = golden("medium_n8192.npz")
    sig = g["sig"]
    for order in (3, 12):
        rows = g[f"rows_o{order}"]
        f, _, cwt = orc.cwt_fft(order, sig, 1000.0)
        assert np.array_equal(f, g[f"f_o{order}"])
        assert relmax(cwt[rows], g[f"cwt_rows_o{order}"]) < 1e-13
        p = np.abs(cwt) ** 2
        assert np.allclose(p.sum(axis=1), g[f"cwt_psum_band_o{order}"], rtol=1e-12)
        assert np.allclose(p.sum(axis=0), g[f"cwt_psum_time_o{order}"], rtol=1e-11)
        assert np.isclose(np.sum(orc.shannon_from_power(p).shannon_bits), float(g[f"cwt_entropy_bits_o{order}"]), rtol=1e-12)
        _, _, stx = orc.stx_fft(order, sig, 1000.0)
        assert relmax(stx[rows], g[f"stx_rows_o{order}"]) < 1e-13
        assert np.isclose((np.abs(stx) ** 2).max(), float(g[f"stx_pmax_o{order}"]), rtol=1e-12)
        c, _, _, fc = orc.cwt_chirp_fft(sig, 1000.0, order)
        assert np.array_equal(fc, g[f"chirp_f_o{order}"])
        assert relmax(c[g[f"chirp_rowsel_o{order}"]], g[f"chirp_rows_o{order}"]) < 1e-10


CASES_STX_GENERAL = {
    "lin": dict(f_min=20.0, f_max=400.0, f_step=20.0),
    "geo": dict(order=3.0, f_min=10.0, f_max=450.0, geometric=True),
    "inferno": dict(order=3.0, f_min=8.0, f_max=400.0, geometric=True, inferno=True),
    "qpr": dict(f_min=25.0, f_max=300.0, f_step=25.0, q=0.5, p=1.0, r=0.75),
}


@pytest.mark.parametrize("name", sorted(CASES_STX_GENERAL))
def test_stx_general(golden, name):
    g = golden("stx_general_n1024.npz")
    tfr, psd, f, f_fft, win = orc.stx_general(g["sig"], 1 / 1000.0, **CASES_STX_GENERAL[name])
    assert np.array_equal(f, g[f"{name}_f"]) and np.array_equal(f_fft, g[f"{name}_ffft"])
    assert relmax(tfr, g[f"{name}_tfr"]) < 1e-14
    assert np.allclose(psd[0], g[f"{name}_psd_row0"], rtol=1e-12)
    assert np.array_equal(win[[0, len(f) - 1]], g[f"{name}_win_rows"])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_welch(golden, dtype):
    g = golden("stft.npz")
    sig = g[f"sig_n13_fs1000_{dtype}"]
    f, p = orc.welch_power_pow2(sig, 1000.0, 512)
    assert np.array_equal(f, g[f"welch_f_{dtype}"]) and p.dtype == g[f"welch_p_{dtype}"].dtype
    assert np.allclose(p, g[f"welch_p_{dtype}"], rtol=1e-12 if dtype == "float64" else 2e-6, atol=0)
    _, p2 = orc.welch_power_pow2(sig, 1000.0, 300, nfft=512, overlap=100, alpha=0.5)
    assert np.allclose(p2, g[f"welch2_p_{dtype}"], rtol=1e-12 if dtype == "float64" else 2e-6, atol=0)


@pytest.mark.parametrize("tag", ["f64", "f32", "f64_odd"])
def test_shannon_1d_family(golden, tag):
    """1-D Shannon TDR / FFT (tfr_info.py:97-200): the oracle against the reference's own outputs, bit for bit."""
    g = golden("shannon1d.npz")
    sig = g[f"sig_{tag}"]
    sig_norm, m_t = orc.shannon_tdr(sig)
    spec, angle, freq, m_f = orc.shannon_fft(sig)
    assert np.array_equal(sig_norm, g[f"tdr_sig_{tag}"]) and np.array_equal(m_t, g[f"tdr_marginal_{tag}"])
    assert np.array_equal(spec, g[f"fft_sig_{tag}"]) and np.array_equal(angle, g[f"fft_angle_{tag}"])
    assert np.array_equal(freq, g[f"fft_frequency_{tag}"]) and np.array_equal(m_f, g[f"fft_marginal_{tag}"])
    for name, m in (("tdr", m_t), ("fft", m_f)):
        info, ent, ref, isnr, esnr = orc.shannon_1d(m)
        assert np.array_equal(info, g[f"{name}_info_{tag}"]) and np.array_equal(ent, g[f"{name}_entropy_{tag}"])
        assert ref == float(g[f"{name}_ref_entropy_{tag}"])
        assert np.array_equal(isnr, g[f"{name}_isnr_{tag}"]) and np.array_equal(esnr, g[f"{name}_esnr_{tag}"])


def _stfft_cases(g):
    for line in g["cases"]:
        tag, fs, alpha, seg, ov, scaling, padding = str(line).split(",")
        yield tag, float(fs), float(alpha), int(seg), int(ov), scaling, padding


def test_short_time_fft_wrappers(golden):
    """utilities/short_time_fft.py (stft_tukey / spectrogram_tukey / istft_tukey over scipy.signal.ShortTimeFFT): the
    oracle's restatement against the reference's outputs."""
    g = golden("short_time_fft.npz")
    sig = g["sig"]
    for tag, fs, alpha, seg, ov, scaling, padding in _stfft_cases(g):
        o = orc.SlidingStft(fs, alpha, seg, ov, scaling)
        assert [o.p_min, o.p_max(len(sig)), o.hop, o.mfft, o.m_mid] == list(g[f"geom_{tag}"])
        f, t, mag = orc.stft_tukey(sig, fs, alpha, seg, ov, scaling, padding)
        assert np.array_equal(f, g[f"f_{tag}"]) and np.array_equal(t, g[f"t_{tag}"])
        assert relmax(mag, g[f"mag_{tag}"]) <= 1e-13
        _, _, sxx = orc.spectrogram_tukey(sig, fs, alpha, seg, ov, scaling, padding)
        assert relmax(sxx, g[f"sxx_{tag}"]) <= 1e-13
        assert relmax(o.stft(sig), g[f"S_{tag}"]) <= 1e-13
        ts, x = orc.istft_tukey(g[f"S_{tag}"], fs, alpha, seg, ov, scaling)
        assert np.array_equal(ts, g[f"ts_{tag}"]) and relmax(x, g[f"x_{tag}"]) <= 1e-12


def test_corner_cases_vs_reference(golden):
    """Gaussian-window STFT, chirped atoms (index_shift = +-1) and the unit-amplitude dictionary against the
    reference's own outputs (tests/golden/corners_n2048.npz)."""
    g = golden("corners_n2048.npz")
    for tag in ("float64", "float32"):
        sig = g[f"sig_{tag}"]
        f, t, z = orc.gtx_complex_pow2(sig, 1000.0, 256)
        assert np.array_equal(f, g[f"gtx_f_{tag}"]) and np.array_equal(t, g[f"gtx_t_{tag}"])
        assert z.dtype == g[f"gtx_z_{tag}"].dtype and np.array_equal(z, g[f"gtx_z_{tag}"])
        f, t, z = orc.gtx_complex_pow2(sig, 1000.0, 200, sigma=30, overlap=150, nfft=512)
        assert np.array_equal(t, g[f"gtx2_t_{tag}"]) and np.array_equal(z, g[f"gtx2_z_{tag}"])
    sig = g["sig_float64"]
    for shift, tag in ((1.0, "p1"), (-1.0, "m1")):
        c, bits, _, fc = orc.cwt_chirp_fft(sig, 1000.0, 3, index_shift=shift)
        assert np.array_equal(fc, g[f"shift_f_{tag}"])
        assert np.array_equal(np.array(orc.chirp_mqg_from_n(3, shift)), g[f"shift_mqg_{tag}"])
        assert relmax(c, g[f"shift_cwt_{tag}"]) < 1e-11
    assert np.max(np.abs(orc.cwt_chirp_fft(sig, 1000.0, 3, index_shift=1.0)[1] - g["shift_bits_p1"])) < 1e-9
    c6 = orc.cwt_chirp_fft(sig, 1000.0, 6, index_shift=1.0, dict_type="spect")[0]
    assert relmax(c6[::3], g["shift_cwt_o6_spect_p1"]) < 1e-11
    f, _, unit = orc.cwt_fft(3, sig, 1000.0, "unit")
    assert np.array_equal(f, g["unit_f_o3"]) and relmax(unit, g["unit_cwt_o3"]) < 1e-14
    atoms = np.stack([orc.gabor_atom_row(3, len(sig), fj, 1000.0, "unit") for fj in f[:4]])
    assert relmax(atoms, g["unit_atoms"]) < 1e-15 and np.array_equal(g["unit_amp"], np.ones(4))


@pytest.mark.parametrize("order,name", [(3, "large_n1048576.npz"), (12, "large_n1048576_o12.npz")])
def test_benchmark_length_rows(golden, order, name):
    """The oracle at the benchmark length (2^20 samples, float32 record) against the reference, band by band for a
    sample of bands of every kind the GPU engines treat differently (lowest = atoms longer than the record, narrow
    spectra, short atoms up to the highest band), at the fixture's ~1250 sampled times; cwt_atoms likewise."""
    g = golden(name)
    n, fs = 1 << 20, 1000.0
    sig = orc.synth_chirp(n, fs, dtype=np.float32)
    assert np.array_equal(sig[:: n // 4096], g["sig_samples"])
    n_b = len(g[f"f_o{order}"])
    pick = sorted({0, 1, n_b // 5, n_b // 2, (3 * n_b) // 4, n_b - 2, n_b - 1})
    tsel = g[f"cwt_tsel_o{order}"]
    f, _, cwt = orc.cwt_fft(order, sig, fs, bands=pick)
    assert np.array_equal(f, g[f"f_o{order}"])
    ref = g[f"cwt_rows_o{order}"][pick]
    assert np.max(np.abs(cwt[:, tsel] - ref)) <= 1e-12 * np.abs(ref).max()
    assert np.allclose((np.abs(cwt) ** 2).sum(axis=1), g[f"cwt_psum_band_o{order}"][pick], rtol=1e-11)
    _, _, stx = orc.stx_fft(order, sig, fs, bands=pick)
    ref = g[f"stx_rows_o{order}"][pick]
    assert np.max(np.abs(stx[:, tsel] - ref)) <= 1e-12 * np.abs(ref).max()
    assert np.allclose((np.abs(stx) ** 2).sum(axis=1), g[f"stx_psum_band_o{order}"][pick], rtol=1e-11)
    n_c = len(g[f"chirp_f_o{order}"])
    pick_c = [0, n_c // 2, n_c - 1]
    c, _, _, fc = orc.cwt_chirp_fft(sig, fs, order, bands=pick_c)
    assert np.array_equal(fc, g[f"chirp_f_o{order}"])
    ref = g[f"chirp_rows_o{order}"][pick_c]
    assert np.max(np.abs(c[:, tsel] - ref)) <= 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("name,log2n,order,dtype,channel,transforms", [
    ("large_n1048576_f64.npz", 20, 3, np.float64, (0, 1), ("cwt", "stx")),
    ("large_n1048576_o12_ch63_stx.npz", 20, 12, np.float32, (63, 64), ("stx",)),
    ("large_n524288_o6.npz", 19, 6, np.float32, (0, 1), ("cwt", "stx")),
    # round 4: the order-12 table on a FLOAT64 record (the shape bench.py's f64 leg and the configs[4] streaming leg time)
    ("large_n1048576_o12_f64.npz", 20, 12, np.float64, (0, 1), ("cwt", "stx")),
])
def test_round3_fixtures_rows(golden, name, log2n, order, dtype, channel, transforms):
    """The oracle against the reference rows added in round 3: the benchmark record in float64 (the reference then works
    in double throughout), channel 63 of the BASELINE configs[2] batch (Stockwell), and an order-6 table at 2^19."""
    g = golden(name)
    n, fs = 1 << log2n, 1000.0
    sig = orc.synth_chirp(n, fs, channel[0], channel[1], dtype=dtype)
    assert np.array_equal(sig[:: n // 4096], g["sig_samples"])
    n_b = len(g[f"f_o{order}"])
    pick = sorted({0, n_b // 3, (2 * n_b) // 3, n_b - 1})
    for name_t, fn in (("cwt", orc.cwt_fft), ("stx", orc.stx_fft)):
        if name_t not in transforms:
            continue
        tsel = g[f"{name_t}_tsel_o{order}"]
        f, _, panel = fn(order, sig, fs, bands=pick)
        assert np.array_equal(f, g[f"f_o{order}"])
        ref = g[f"{name_t}_rows_o{order}"][pick]
        assert np.max(np.abs(panel[:, tsel] - ref)) <= 1e-12 * np.abs(ref).max()
        assert np.allclose((np.abs(panel) ** 2).sum(axis=1), g[f"{name_t}_psum_band_o{order}"][pick], rtol=1e-11)
