"""CPU-only checks of the host side of the product: band / index tables bit-exact against the
golden vectors, windows against SciPy, the C-ABI library loads and exports every symbol the header
declares, and the transforms refuse to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import quantum_inferno_amd as qi
from quantum_inferno_amd import _lib, cwt_atoms, scales_dyadic as sd, styx_fft
from quantum_inferno_amd.utilities import calculations, matrix, rescaling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_known_answer_of_reference_tests():
    # reference tests/test_scales_dyadic.py:8-21 (commented out upstream)
    f = sd.log_frequency_hz_from_fft_points(100.0, 8192, 6, 1.0, sd.Slice.G3)
    assert len(f) == 48 and f[0] == 0.1778279410038923 and f[-1] == 39.810717055349706
    # reference tests/utilities/test_rescaling.py:7-21, test_calculations.py:86-100
    assert round(float(rescaling.to_log2_with_epsilon(100.0)), 2) == 6.64
    assert round(float(rescaling.to_log2_with_epsilon(-100.0)), 2) == 6.64
    import torch

    host = rescaling.to_log2_with_epsilon(torch.tensor([100.0, -3.0], dtype=torch.float32))  # host tensor -> NumPy float64, as the
    assert isinstance(host, np.ndarray) and host.dtype == np.float64                         # signature promises (rescaling.py:13-20)
    assert np.array_equal(host, np.log2(np.abs(np.array([100.0, -3.0])) + sd.get_epsilon()))
    assert rescaling.is_power_of_two(8) and not rescaling.is_power_of_two(9)
    assert calculations.round_value(1.5, "round") == 2
    assert calculations.get_num_points(10, 10, "round", "log2") == 7
    with pytest.raises(ValueError):
        calculations.round_value(1.5, "nearest")
    with pytest.raises(ValueError):
        calculations.get_num_points(10, 10, "round", "bits")


def test_band_tables_bit_exact(golden):
    g = golden("bands.npz")
    for key in g["combos"]:
        fs_s, n_s, o_s = str(key).split("_")
        fs, n, order = float(fs_s[2:]), 2 ** int(n_s[1:]), int(o_s[1:])
        f = sd.log_frequency_hz_from_fft_points(fs, n, order)
        assert np.array_equal(f, g[f"f_{key}"]), key
        assert np.array_equal(sd.stx_shift_indices(f, n, fs), g[f"idx_{key}"]), key
        assert np.array_equal(sd.scale_from_frequency_hz(order, f, fs)[0], g[f"scale_{key}"]), key
        _, f_min = cwt_atoms.chirp_scales_from_duration(order, n / fs)
        out = cwt_atoms.chirp_frequency_bands(order, f_min, fs, fs / 2.0)
        assert np.array_equal(np.flip(out[4]), g[f"chirpf_{key}"]), key
        assert np.array_equal(np.array(out[:4], dtype=np.float64), g[f"chirpmq_{key}"]), key


def test_scalars_bit_exact(golden):
    g = golden("bands.npz")
    for fs, order, seg in g["stft_seg"]:
        assert styx_fft.stft_segment_points(fs, order) == int(seg)
    with pytest.warns(UserWarning):
        assert sd.cycles_from_order(0.5) == g["cycles"][0]
    assert np.array_equal(np.array([sd.cycles_from_order(o) for o in (0.75, 1, 3, 6, 12, 24)]), g["cycles"][1:])
    assert np.array_equal(np.array([cwt_atoms.chirp_mqg_from_n(o) for o in (1, 3, 6, 12, 24)]), g["mqg"])
    assert sd.get_epsilon() == 2.220446049250313e-16


def test_windows_match_scipy():
    import scipy.signal as ss

    for m in (64, 256, 512, 2048, 1000):
        for alpha in (0.0, 0.25, 0.5, 1.0):
            assert np.array_equal(styx_fft.tukey_window_periodic(m, alpha), ss.get_window(("tukey", alpha), m)), (m, alpha)
        assert np.array_equal(styx_fft.gaussian_window_periodic(m, m // 4), ss.get_window(("gaussian", m // 4), m))


def test_tile_multiplies():
    a = np.arange(12.0).reshape(3, 4)
    assert np.array_equal(matrix.d0tile_x_d0d1(np.array([1.0, 2.0, 3.0]), a), np.array([1.0, 2.0, 3.0])[:, None] * a)
    assert np.array_equal(matrix.d1tile_x_d0d1(np.arange(4.0), a), np.arange(4.0)[None, :] * a)
    with pytest.raises(TypeError):
        matrix.d0tile_x_d0d1(np.arange(4.0), a)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "qi_tfr.h")).read()
    declared = set(re.findall(r"\b(qi_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    assert lib.qi_abi_version() == 1
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/qi_tfr.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    # pure-host entry points may be called without a GPU
    assert lib.qi_stft_segments(65536, 512, 256) == 257
    assert lib.qi_stft_segments(65536, 2048, 1024) == 65
    assert lib.qi_stft_segments(8192, 512, 256) == 33


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.QiError):
        qi.styx_stx.stx_complex_any_scale_pow2(3, np.zeros(1024), 1000.0)
    with pytest.raises(_lib.QiError):
        qi.styx_cwt.cwt_complex_any_scale_pow2(3, np.zeros(1024), 1000.0)
    with pytest.raises(_lib.QiError):
        qi.tfr_info.scale_power_bits(np.ones((4, 16)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "quantum_inferno_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), f"{fn} mentions the oracle"


def test_stream_chunking_cursor():
    from quantum_inferno_amd import stream

    # BASELINE config 5: 24 h at 800 Hz in chunks of 2^20 with a hop of 2^19
    starts = stream.chunk_starts(69_120_000, 1 << 20, 1 << 19)
    assert len(starts) == 131 and starts[0] == 0 and starts[1] == 1 << 19
    assert starts[-1] + (1 << 20) == 69_120_000 and np.all(np.diff(starts) > 0)
    assert np.all(np.diff(starts)[:-1] == 1 << 19)
    covered = np.zeros(69_120_000 // 4096, dtype=bool)  # 69 120 000 = 16 875 * 4096
    for s0 in starts:
        covered[s0 // 4096 : (s0 + (1 << 20) + 4095) // 4096] = True
    assert covered.all()
    assert np.array_equal(stream.chunk_starts(4096, 1024, 1024), [0, 1024, 2048, 3072])
    with pytest.raises(ValueError):
        stream.chunk_starts(1000, 1024, 512)
    x = np.arange(2 * 5000, dtype=np.float64).reshape(2, 5000)
    got = [(i, s0, v.shape) for i, s0, v in stream.iter_chunks(x, 2048, 1024, first_chunk=1)]
    assert got == [(1, 1024, (2, 2048)), (2, 2048, (2, 2048)), (3, 2952, (2, 2048))]


def test_sliding_stft_geometry_matches_scipy():
    """The host restatement of scipy.signal.ShortTimeFFT's slice geometry and canonical dual window
    (quantum_inferno_amd/utilities/short_time_fft.py: TukeyStft) against SciPy itself over a sweep of shapes."""
    import scipy.signal as ss

    from quantum_inferno_amd.utilities import short_time_fft as stf

    for seg, overlap, alpha, scaling in [(256, 128, 0.25, "magnitude"), (200, 150, 0.5, "psd"), (128, 96, 1.0, "magnitude"),
                                         (64, 1, 0.0, "magnitude"), (33, 11, 0.3, "psd"), (500, 499, 0.25, "magnitude"),
                                         (16, 8, 0.25, None)]:
        obj = stf.get_stft_object_tukey(800.0, alpha, seg, overlap, scaling)
        win = ss.windows.tukey(seg, alpha=alpha)
        np.testing.assert_array_equal(stf.tukey_window_symmetric(seg, alpha), win)
        ref = ss.ShortTimeFFT(win=win, hop=seg - overlap, fs=800.0, mfft=obj.mfft, fft_mode="onesided", scale_to=scaling)
        np.testing.assert_array_equal(obj.win, ref.win)
        np.testing.assert_array_equal(obj.f, ref.f)
        assert (obj.p_min, obj.k_min, obj.m_num_mid, obj.delta_t) == (ref.p_min, ref.k_min, ref.m_num_mid, ref.delta_t)
        for n in (seg, seg + 1, 3 * seg + 7, 4096, 10007):
            assert obj.p_max(n) == ref.p_max(n) and obj.k_max(n) == ref.k_max(n), (seg, overlap, n)
        np.testing.assert_array_equal(obj.dual_win, ref.dual_win)


def test_stream_work_items_and_rank_shares():
    """Config 5's work list: (channel block, chunk) items of the 24 h x 1024-channel job, dealt to 8 ranks in contiguous
    shares that cover every item exactly once (no GPU needed: pure index arithmetic)."""
    from quantum_inferno_amd import stream

    items = stream.work_items(1024, 16, 69_120_000, 1 << 20, 1 << 19)
    starts = stream.chunk_starts(69_120_000, 1 << 20, 1 << 19)
    assert len(starts) == 131 and len(items) == 64 * 131
    assert items[0] == (0, 16, 0, 0) and items[131] == (16, 16, 0, 0) and items[130][3] == 69_120_000 - (1 << 20)
    seen = []
    for r in range(8):
        share = stream.rank_items(items, r, 8)
        assert len(share) == 8 * 131  # eight blocks of 16 channels per GPU = its 128 channels
        seen += share
    assert seen == items
    ragged = stream.work_items(5, 2, 3000, 1024, 512)
    assert [it[:2] for it in ragged[::len(stream.chunk_starts(3000, 1024, 512))]] == [(0, 2), (2, 2), (4, 1)]
    assert sum(len(stream.rank_items(ragged, r, 3)) for r in range(3)) == len(ragged)
