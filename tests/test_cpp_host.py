"""C++ host mirror (zen_amd/libzen) and the `zen` command line tool.

CPU part: the binaries build, `zen version` / `zen help` work and `--cpu` is refused.
GPU part: tests/cpp/test_libzen (the reference's gtest cases against the C++ classes, oracle as checker)
and an end-to-end run of `zen offline` / `zen fakert` on a synthetic WAV whose PCM16 outputs are compared
sample for sample with what the reference CLI's conversions (zen/offline.h:180-223, zen/fakert.h:15-34,
:117-287) yield when fed the oracle's waveforms."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZEN = os.path.join(ROOT, "zen_amd", "bin", "zen")
TEST_EXE = os.path.join(ROOT, "tests", "cpp", "test_libzen")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.path.exists(ZEN) and os.path.exists(TEST_EXE)):
        from oracle import oracle as o
        from zen_amd import build
        o.build()
        build.build()
        build.build_host()


def write_wav_pcm16(path, x, fs, channels=1):
    pcm = np.asarray(x, dtype="<i2")
    data = pcm.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 1, channels, fs, fs * 2 * channels, 2 * channels, 16) +
                b"data" + struct.pack("<I", len(data)) + data)


def read_wav_pcm16(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE" and b[12:16] == b"fmt " and b[36:40] == b"data"
    fmt, ch, fs, _, _, bits = struct.unpack("<HHIIHH", b[20:36])
    assert (fmt, ch, bits) == (1, 1, 16)
    n = struct.unpack("<I", b[40:44])[0]
    return fs, np.frombuffer(b[44:44 + n], dtype="<i2")


def test_cli_version_help_and_cpu_refusal(tmp_path):
    assert subprocess.run([ZEN, "version"], capture_output=True, text=True).stdout == "version 1.0\n"
    out = subprocess.run([ZEN, "help"], capture_output=True, text=True).stdout
    assert "zen offline" in out and "zen fakert" in out and "--only-percussive" in out
    r = subprocess.run([ZEN, "offline", "-i", "x.wav", "--cpu"], capture_output=True, text=True)
    assert r.returncode == 2 and "--cpu" in r.stderr


@pytest.mark.gpu
def test_cpp_reference_suites():
    r = subprocess.run([TEST_EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout


def pcm16(x):
    v = (x.astype(np.float32) * np.float32(32767.0)).astype(np.float64)   # float32 product, exact in f64
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int64)      # lroundf: half away from zero


@pytest.mark.gpu
def test_cli_offline_and_fakert_end_to_end(tmp_path):
    from oracle import oracle as o
    fs, n = 44100, 161571                      # README.md:100 clip length
    rng = np.random.default_rng(0)
    t = np.arange(n) / fs
    left = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.uniform(-1, 1, n)
    right = 0.3 * np.sin(2 * np.pi * 660 * t) + 0.1 * rng.uniform(-1, 1, n)
    left[::11025] += 0.5
    st = np.empty(2 * n, np.int16)
    st[0::2] = np.round(left * 20000)
    st[1::2] = np.round(right * 20000)
    wav = str(tmp_path / "in.wav")
    write_wav_pcm16(wav, st, fs, channels=2)
    # what the CLI does to the file: /32767.f then (L+R)/2.0f
    f32 = st.astype(np.float32) / np.float32(32767.0)
    mono = ((f32[0::2] + f32[1::2]) / np.float32(2.0)).astype(np.float32)

    r = subprocess.run([ZEN, "offline", "-i", wav, "--hps", "4096", "2.5", "256", "2.5", "-o", str(tmp_path / "off")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Processing input signal of size 161571 with HPR-I separation using harmonic params: 4096,2.5, " \
           "percussive params: 256,2.5" in r.stdout
    assert "2-pass HPR-I-Offline took" in r.stdout and "compute: gpu" in r.stdout
    h, p, _ = o.HPRIOffline(float(fs), 4096, 256, 2.5, 2.5).process(mono)
    for name, ref in (("off_harm.wav", h), ("off_perc.wav", p)):
        rfs, got = read_wav_pcm16(str(tmp_path / name))
        peak = max(-ref.min(), ref.max())
        exp = pcm16(ref / np.float32(peak))
        assert rfs == fs and np.array_equal(got.astype(np.int64), exp), name
    assert os.path.exists(str(tmp_path / "off_residual.wav"))     # all-zero / max 0 -> NaN -> written as is

    hop = 1024
    r = subprocess.run([ZEN, "fakert", "-i", wav, "--hps", str(hop), "2.0", "-o", str(tmp_path / "rt.wav")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    n_chunks = len(range(0, n - hop, hop))                        # fakert.h:23-27
    assert "Slicing buffer size %d into %d chunks of size %d" % (n, n_chunks, hop) in r.stdout
    assert "average processing duration(us)" in r.stdout
    eng = o.HPR(float(fs), hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL)
    ref = mono.copy()                                             # fakert.h:132: output starts as the input
    ref[:n_chunks * hop] = eng.process_stream(mono[:n_chunks * hop])["P"]
    peak = max(-ref.min(), ref.max())
    _, got = read_wav_pcm16(str(tmp_path / "rt.wav"))
    assert np.array_equal(got.astype(np.int64), pcm16(ref / np.float32(peak)))
    # the same stream through the resident kernel (MI355X extension, HPRRealtime::use_resident_kernel): the same file
    r = subprocess.run([ZEN, "fakert", "-i", wav, "--hps", str(hop), "2.0", "-o", str(tmp_path / "rt_res.wav"), "--resident", "50"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "average processing duration(us)" in r.stdout
    _, got_res = read_wav_pcm16(str(tmp_path / "rt_res.wav"))
    assert np.array_equal(got_res, got)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [("--sse",), ("--soft-mask",), ("--nocopybord",), ("--soft-mask", "--nocopybord"),
                                   ("--only-percussive",)])
def test_cli_mask_filter_and_border_flags(tmp_path, flags):
    """The option switches of the reference CLI (zen/main.cpp: --sse, --soft-mask, --nocopybord,
    --only-percussive; zen/offline.h:150-163, :196-258 and zen/fakert.h:88-100): each reaches the engine the way
    use_sse_filter() / use_soft_mask() / the nocopybord constructor argument do, and --only-percussive writes the
    percussive file alone."""
    from oracle import oracle as o
    fs, n = 22050, 52000
    rng = np.random.default_rng(11)
    t = np.arange(n) / fs
    x = 0.4 * np.sin(2 * np.pi * 523.25 * t) + 0.2 * rng.uniform(-1, 1, n) * (np.arange(n) % 5000 < 200)
    pcm = np.round(x * 20000).astype(np.int16)
    wav = str(tmp_path / "in.wav")
    write_wav_pcm16(wav, pcm, fs)
    mono = (pcm.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    sse, soft, nocopy = "--sse" in flags, "--soft-mask" in flags, "--nocopybord" in flags

    r = subprocess.run([ZEN, "offline", "-i", wav, "--hps", "1024", "2.0", "256", "2.0", "-o", str(tmp_path / "off")]
                       + list(flags), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert ("mask: soft/Wiener" if soft else "mask: hard/binary") in r.stdout
    assert ("filter: sse" if sse else "filter: median") in r.stdout
    ref = o.HPRIOffline(float(fs), 1024, 256, 2.0, 2.0, nocopybord=nocopy)
    if sse:
        ref.use_sse_filter()
    if soft:
        ref.use_soft_mask()
    h, p, _ = ref.process(mono)
    wanted = (("off_perc.wav", p),) if "--only-percussive" in flags else (("off_harm.wav", h), ("off_perc.wav", p))
    for name, w in wanted:
        rfs, got = read_wav_pcm16(str(tmp_path / name))
        peak = max(-w.min(), w.max())
        assert rfs == fs and np.array_equal(got.astype(np.int64), pcm16(w / np.float32(peak))), (flags, name)
    if "--only-percussive" in flags:
        assert sorted(f for f in os.listdir(str(tmp_path)) if f.startswith("off")) == ["off_perc.wav"]
        return                                                    # fakert has no such switch

    hop = 512
    r = subprocess.run([ZEN, "fakert", "-i", wav, "--hps", str(hop), "2.0", "-o", str(tmp_path / "rt.wav")]
                       + list(flags), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    n_chunks = len(range(0, n - hop, hop))
    eng = o.HPR(float(fs), hop, 2.0, o.OUTPUT_PERCUSSIVE, o.TIME_CAUSAL, copy_bord=not nocopy)
    if sse:
        eng.use_sse_filter()
    if soft:
        eng.use_soft_mask()
    w = mono.copy()
    w[:n_chunks * hop] = eng.process_stream(mono[:n_chunks * hop])["P"]
    peak = max(-w.min(), w.max())
    _, got = read_wav_pcm16(str(tmp_path / "rt.wav"))
    assert np.array_equal(got.astype(np.int64), pcm16(w / np.float32(peak))), flags


@pytest.mark.gpu
def test_cli_batch_directory(tmp_path):
    """`zen batch`: a directory of clips (two lengths) separated in batches == per-clip `zen offline`."""
    fs = 44100
    rng = np.random.default_rng(3)
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    outdir.mkdir()
    names = []
    for i, n in enumerate((30000, 30000, 30000, 41000)):
        x = np.round(rng.uniform(-1, 1, n) * 12000).astype(np.int16)
        write_wav_pcm16(str(indir / ("clip%d.wav" % i)), x, fs)
        names.append("clip%d" % i)
    r = subprocess.run([ZEN, "batch", "-i", str(indir), "-o", str(outdir), "--hps", "1024", "2.0", "256", "2.0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "4 wav files in 2 (rate, length) groups" in r.stdout
    for nm in names:
        r1 = subprocess.run([ZEN, "offline", "-i", str(indir / (nm + ".wav")), "--hps", "1024", "2.0", "256", "2.0",
                             "-o", str(tmp_path / nm)], capture_output=True, text=True, timeout=600)
        assert r1.returncode == 0, r1.stderr
        for suffix in ("_harm.wav", "_perc.wav"):
            a = open(str(outdir / (nm + suffix)), "rb").read()
            b = open(str(tmp_path / (nm + suffix)), "rb").read()
            assert a == b, nm + suffix


def test_median_networks_on_the_host(tmp_path):
    """zen_amd/csrc/median_net.h compiled as plain C++ (tests/cpp/test_median_net_host.cpp): every sorting /
    selection network the kernels instantiate, and the 47-tap block scheme of median47_dpp_kernel with its
    neighbours' sorted pieces, against a brute-force replicate-border median (mfilt.h:270-342 semantics)."""
    exe = str(tmp_path / "test_median_net_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_median_net_host.cpp"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "passed" in r.stdout, r.stdout[-2000:]


def test_block_merge_selection_on_the_host(tmp_path):
    """zen_amd/csrc/median_big.h compiled as plain C++ (tests/cpp/test_median_big_host.cpp): zbig::medians_big<W> for every long
    mask the kernels instantiate (65 .. 257 taps), through a loader of sorted 16-blocks and through one that also hands out
    the sorted PAIRS of blocks neighbouring threads share (round 6, median_big.hip), against a brute-force median."""
    exe = str(tmp_path / "test_median_big_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_median_big_host.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 failures" in r.stdout and "W = 257" in r.stdout, r.stdout[-2000:]


def test_real_input_transform_on_the_host(tmp_path, oracle):
    """zen_amd/csrc/rfft_dev.h compiled for the CPU (tests/cpp/test_rfft_host.cpp; the image's host clang++, scalar
    butterflies): the Hermitian half of the radix-2 DAG on real frames, the N/32 threads of a frame run one after the other,
    bit-identical to the oracle's complex transform (fftw.h:51-129; the analysis of hps.cu:456-465) for nfft 32..16384,
    zero-padded and not, every bin 0..nfft/2 delivered exactly once."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no host clang++ (the header uses clang vector extensions)")
    exe = str(tmp_path / "test_rfft_host")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call([clang, "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-everything", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "test_rfft_host.cpp"), "-o", exe, "-L", odir, "-lzen_oracle",
                           "-Wl,-rpath," + odir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-2000:]


def test_latency_layout_transform_on_the_host(tmp_path, oracle):
    """zen_amd/csrc/lfft_dev.h compiled for the CPU (tests/cpp/test_lfft_host.cpp): the transform of the single-hop kernels --
    4, 8 or 16 values per thread, two LDS images, one barrier per pass -- run thread by thread in lock step and one thread
    after the other (only legal because a pass writes the image it does not read): bit-identical to the oracle's transform
    (fftw.h:51-129) forward (zero-padded and full frames) and inverse (all samples, first half only) for nfft 64..8192."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no host clang++ (the header uses clang vector extensions)")
    exe = str(tmp_path / "test_lfft_host")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call([clang, "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-everything", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "test_lfft_host.cpp"), "-o", exe, "-L", odir, "-lzen_oracle",
                           "-Wl,-rpath," + odir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_cli_batch_over_two_processes_matches_one(tmp_path):
    """`zen batch --gpus 2` (SURVEY 8(e), configs[3]): the parent starts two copies of itself before anything has
    touched a GPU, files are dealt round robin, every child separates its own.  Same PCM16 files as one process."""
    rng = np.random.default_rng(3)
    ind, out1, out2 = tmp_path / "in", tmp_path / "o1", tmp_path / "o2"
    for d in (ind, out1, out2):
        d.mkdir()
    for i in range(5):
        n = 30000 if i != 3 else 41000               # two (rate, length) groups
        x = (0.4 * np.sin(2 * np.pi * 330 * np.arange(n) / 44100.0) + 0.3 * rng.uniform(-1, 1, n) * (np.arange(n) % 4000 < 100))
        write_wav_pcm16(str(ind / ("clip%d.wav" % i)), np.round(x * 20000).astype(np.int16), 44100)
    args = [ZEN, "batch", "-i", str(ind), "--hps", "1024", "2.0", "256", "2.0"]
    r1 = subprocess.run(args + ["-o", str(out1)], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    env = dict(os.environ, ZEN_ALLOW_GPU_SHARING="1")  # one GPU here: both children use device 0
    r2 = subprocess.run(args + ["-o", str(out2), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "zen batch: 2 GPUs: 5 files" in r2.stdout and "[gpu 0]" in r2.stdout and "[gpu 1]" in r2.stdout
    names = sorted(os.listdir(out1))
    assert names == sorted(os.listdir(out2)) and len(names) == 10
    for nme in names:
        assert open(os.path.join(out1, nme), "rb").read() == open(os.path.join(out2, nme), "rb").read(), nme
