"""The multi-GPU host, rehearsed in C on however many GPUs the box has: tests/native/multi_handle_host.c opens N model handles
from N threads (N = 8 here; on a one-GPU box all eight share device 0) BY MODEL SPEC "model.k2w@device" -- the naming the C# shim
parses out of the reference's unchanged constructors -- and drives them through native OfflineStreams (AddSamples in two pieces,
GetResults: the calls csharp/OfflineRecognizer.Hip.cs makes), cuts BASELINE configs[2]'s utterance list -- 256 x 10 s, modified beam
search beam 4 -- and configs[4]'s -- 64 x 30 s conformer-zh, 8 per GPU -- into the shards and GetResults batches of
k2transducerasr_amd/shard.py, and requires the concatenation to equal the same list decoded through ONE handle.  That is the C# host's exact shape (one OfflineRecognizer per GPU,
one thread each, INTEGRATION.md "More than one GPU"); no 1 -> 8 GPU scaling curve exists yet (the driver's 8-GPU node was never
available), so this is what proves the sharded path before it meets one."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("mh") / "multi_handle_host")
    lib = os.path.join(ROOT, "k2transducerasr_amd")
    subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "multi_handle_host.c"), "-o", exe, "-L", lib, "-lk2hip", "-Wl,-rpath," + lib], check=True)
    return exe


def read_results(path):
    raw = open(path, "rb").read()
    total, mt = np.frombuffer(raw, np.int32, 2)
    off, out = 8, []
    for _ in range(total):
        n = int(np.frombuffer(raw, np.int32, 1, off)[0])
        tok = np.frombuffer(raw, np.int64, mt, off + 4)
        ts = np.frombuffer(raw, np.int32, mt, off + 4 + 8 * mt)
        out.append((tok[:n].tolist(), ts[:n].tolist()))
        off += 4 + 12 * mt
    return out


@pytest.mark.parametrize("preset,total,secs,batch,handles,beam", [
    ("zipformer2-tiny-test", 13, 1.1, 3, 4, 0),              # ragged shards (13 over 4 handles: 4 / 3 / 3 / 3), a short last batch
    ("zipformer2-large-en", 256, 10.0, 32, 8, 4),           # BASELINE configs[2]
    ("conformer-zh", 64, 30.0, 8, 8, 0),                    # BASELINE configs[4]: 64 x 30 s, 8 per GPU (round 5)
])
def test_n_handles_from_n_threads_equal_one_handle(host_exe, tmp_path, preset, total, secs, batch, handles, beam):
    from k2transducerasr_amd import Model
    from k2transducerasr_amd.shard import batches_of, shard_range
    from k2transducerasr_amd.synth import synth_utterance, write_synthetic_model
    model = str(tmp_path / "m.k2w")
    write_synthetic_model(model, preset)
    s = np.stack([synth_utterance(u, secs) for u in range(total)]).astype(np.float32)
    samples = str(tmp_path / "samples.f32")
    s.tofile(samples)
    out = str(tmp_path / "out.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([host_exe, model, samples, str(total), str(s.shape[1]), str(batch), str(handles), str(beam), out],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, (p.stdout, p.stderr)
    print(p.stdout.strip())
    got = read_results(out)
    assert len(got) == total and sum(len(t) for t, _ in got) > 0
    # ... and the C host's shards are the Python package's: the first and the last handle's batches through the ctypes binding
    m = Model(model, 0)
    if beam:
        m.set_decoding_method("modified_beam_search", beam)
    for rank in (0, handles - 1):
        lo, hi = shard_range(total, handles, rank)
        for first, cnt in batches_of(lo, hi, batch)[:1]:
            assert m.offline_greedy_from_samples(list(s[first: first + cnt])) == got[first: first + cnt], (rank, first)
    m.close()


def test_streaming_handles_on_host_threads(tmp_path):
    """Four OnlineRecognizers, four host threads, one GPU.  The library keeps all of its blocking transfers on a utility stream of its
    own per device (csrc/common.h copy_blocking; found in round 4 by running this shape); here every thread walks through changing
    stream counts, creates and closes streams, reads states and synchronises while the others do the same, and every stream's tokens
    must equal the same stream decoded by one handle alone."""
    # the scenario lives in tests/streaming_threads_child.py and runs as a child process: a fault of the runtime under five threads of
    # allocations and frees would otherwise end the whole test session instead of failing this test
    import sys
    child = os.path.join(ROOT, "tests", "streaming_threads_child.py")
    r = subprocess.run([sys.executable, child, str(tmp_path / "s.k2w")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(r.stdout.strip())
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.strip().startswith("threads ok:")


def test_foreign_legacy_stream_traffic_beside_the_ticks(tmp_path):
    """Another thread of the HOST process (not this library) keeps issuing legacy-stream copies -- what a plain hipMemcpy or a
    framework on the default stream does -- while a recognizer decodes through changing stream counts.  The library's streams are
    non-blocking and it never captures one (the hipGraph replay of rounds 3 - 4 made the runtime refuse such copies while a tick was
    being recorded; it measured equal to eager enqueue and was removed in round 5), so: no copy is refused, the tokens equal an
    undisturbed recognizer's, and a crash of the child is a FAILURE, not a skip."""
    import sys
    child = os.path.join(ROOT, "tests", "foreign_legacy_child.py")
    r = subprocess.run([sys.executable, child, str(tmp_path / "s.k2w")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(r.stdout.strip())
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "tokens equal: True" in r.stdout and "refused by the runtime: 0" in r.stdout
