"""CPU sanitizer build (AddressSanitizer + UBSan) of libk2hip's host code: the units that read untrusted bytes -- the .k2w container
parser (csrc/k2w_file.cpp) and the token -> text stage (csrc/text.cpp) -- and, over a CPU stand-in of the engine
(tests/native/engine_stub.cpp), the host layer itself: csrc/api.cpp's OfflineStream / OnlineStream mirrors (feature FIFO, lazy fbank,
RemoveChunk, IsFinished, device-mirror bookkeeping, poisoning after a failed step), argument checks and error transport, plus the
real config parser of csrc/model.cpp.  GPU sanitizers do not exist on this pool, and this is
where a malformed input could hurt: a truncated, corrupted or mismatched file must come back as K2HIP_ERR_IO, never as a crash
(the reference's contract: every failure of the operator is an exception, OfflineProjOfTransducer.cs:87-90)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "native", "k2hip_san_driver")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def driver():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "k2transducerasr_amd", "csrc"), "-s", "san"])
    return DRIVER


def run(driver, *args):
    r = subprocess.run([driver, *args], capture_output=True, text=True, env=ENV, timeout=300)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    return r.stdout.strip()


def test_parser_accepts_the_writer_and_rejects_damage(driver, tiny_model_path, tmp_path):
    out = run(driver, "k2w", tiny_model_path)
    assert out.startswith("OK ")
    raw = open(tiny_model_path, "rb").read()
    (data_off,) = struct.unpack_from("<Q", raw, 16)
    cases = {
        "empty": b"",
        "magic": b"K2W2" + raw[4:],
        "version": raw[:4] + struct.pack("<I", 2) + raw[8:],
        "short_header": raw[:20],
        "cut_in_metadata": raw[:100],
        "cut_in_table": raw[: data_off - 40],
        "cut_in_data": raw[: data_off + 1000],
        "data_off_past_eof": raw[:16] + struct.pack("<Q", len(raw) + 64) + raw[24:],
        "data_off_wraps": raw[:16] + struct.pack("<Q", 2**64 - 8) + raw[24:],
        "huge_counts": raw[:8] + struct.pack("<II", 2**31, 2**31) + raw[16:],
    }
    for name, blob in cases.items():
        p = tmp_path / f"{name}.k2w"
        p.write_bytes(blob)
        assert run(driver, "k2w", str(p)).startswith("ERR -2 "), name     # K2HIP_ERR_IO
    assert run(driver, "k2w", str(tmp_path / "missing.k2w")).startswith("ERR -2 ")


def test_tensor_records_are_range_checked(driver, tmp_path):
    from k2transducerasr_amd.k2w import write_k2w
    p = str(tmp_path / "m.k2w")
    write_k2w(p, {"model_type": "zipformer2"}, [("a.weight", np.arange(12, dtype=np.float32).reshape(3, 4)), ("b.idx", np.arange(5, dtype=np.int64))])
    assert run(driver, "k2w", p).startswith("OK 1 2")
    raw = bytearray(open(p, "rb").read())
    at = raw.index(b"a.weight") + len(b"a.weight")          # -> dtype, ndim, dims[4], off, nbytes
    def patched(off, fmt, val):
        b = bytearray(raw)
        struct.pack_into(fmt, b, at + off, val)
        q = tmp_path / "x.k2w"
        q.write_bytes(bytes(b))
        return run(driver, "k2w", str(q))
    assert patched(0, "<I", 7).startswith("ERR -2")          # unknown dtype
    assert patched(4, "<I", 5).startswith("ERR -2")          # ndim > 4
    assert patched(8, "<Q", 2**62).startswith("ERR -2")      # dims overflow
    assert patched(8, "<Q", 4).startswith("ERR -2")          # nbytes != numel * 4
    assert patched(40, "<Q", 2**64 - 16).startswith("ERR -2")  # offset + nbytes wraps
    assert patched(40, "<Q", 2**20).startswith("ERR -2")     # offset past the file
    assert patched(40, "<Q", 2).startswith("ERR -2")         # misaligned
    assert patched(48, "<Q", 2**40).startswith("ERR -2")     # nbytes too large


def test_fuzzed_containers_never_crash(driver, tiny_model_path, tmp_path):
    import shutil
    p = str(tmp_path / "f.k2w")
    shutil.copy(tiny_model_path, p)
    out = run(driver, "fuzz", p, "20240607", "400")
    ok, err, other = (int(x.split("=")[1]) for x in out.split()[1:])
    assert other == 0 and err > 100 and ok + err == 400, out


def test_text_stage_under_sanitizers(driver, tmp_path):
    p = tmp_path / "tokens.txt"
    p.write_text("<blk> 0\n<sos/eos> 1\n<unk> 2\n▁HE 3\nLLO 4\n<0xE4> 5\n<0xBD> 6\n<0xA0> 7\n▁ 8\n<0xZZ> 9\n<0xE 10\n", encoding="utf-8")
    assert run(driver, "text", str(p), "0", "3", "4") == "hello"
    assert run(driver, "text", str(p), "0", "-1", "0", "5", "6", "7").endswith("你")
    assert run(driver, "text", str(p), "0", "5", "6").strip() != ""             # an incomplete UTF-8 run: U+FFFD, no crash
    run(driver, "text", str(p), "0", "9", "10", "8")                            # malformed hex tokens
    assert run(driver, "text", str(p), "0", "99").startswith("ERR -1")          # id outside tokens.txt -> K2HIP_ERR_INVALID
    assert run(driver, "text", str(tmp_path / "none.txt"), "0", "1").startswith("ERR -2")


# ---- the host layer (csrc/api.cpp) under the sanitizers, over tests/native/engine_stub.cpp ---------------------------------------
API_DRIVER = os.path.join(ROOT, "tests", "native", "k2hip_san_api_driver")


@pytest.fixture(scope="module")
def api_driver(driver):
    return API_DRIVER


@pytest.fixture(scope="module")
def streaming_tiny_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("san") / "stream.k2w")
    write_synthetic_model(p, "zipformer2-streaming-tiny-test")
    return p


def test_api_error_paths_poisoning_and_mirror_recovery(api_driver, streaming_tiny_path):
    """Null arguments, a stream of another model, one stream twice in a GetResults list, a device failure inside a chunk step (nothing
    host-side moves, the streams refuse further steps until reset), the same for the operator-level states, and the FIFO mirror's
    return after an oversized block -- all through the C ABI, api.cpp compiled with ASan + UBSan (OnlineStream.cs:82-161,
    IOnlineProj.cs:65-71)."""
    assert run(api_driver, "errors", streaming_tiny_path) == "errors ok"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_online_stream_bookkeeping_fuzz(api_driver, streaming_tiny_path, seed):
    """4 000 random AddSamples (ragged lengths, empty pushes) / AddFeatures / GetResults over random subsets / IsFinished / Reset /
    destroy-and-create calls over 7 streams; after every call SpeechLength, the decodable flag of every stream and the token counts
    must equal a plain model of OnlineStream.cs:57-161 kept by the driver."""
    out = run(api_driver, "online", streaming_tiny_path, str(seed), "4000")
    assert out.startswith("online ok: 4000 rounds")
    assert int(out.split(",")[1].split()[0]) > 1000          # chunk steps actually ran


def test_offline_stream_and_pipeline_paths(api_driver, tiny_model_path):
    """OfflineStream AddSamples in pieces, GetResult / GetResults, submit / wait with exactly-sized output buffers, a third submit and a
    second wait refused, decoder / joiner operators (OfflineStream.cs:43-68, OfflineRecognizer.cs:77-91)."""
    out = run(api_driver, "offline", tiny_model_path, "7", "60")
    assert out.startswith("offline ok: 60 rounds")


def test_model_type_is_derived_as_the_reference_derives_it(api_driver, tiny_model_path, streaming_tiny_path, tmp_path):
    """CustomMetadata.Model_type as OfflineModel.cs:51-63 / OnlineModel.cs:96-106 compute it and OfflineRecognizer.cs:38-53 /
    OnlineRecognizer.cs:26-44 route it, through the real csrc/model.cpp (k2hip_model_create + k2hip_model_meta):
      * offline: a comment whose lower-cased text holds "ctc" and "zipformer2" makes a model_type "zipformer2" container a
        zipformer2ctc one (how icefall's CTC exports look); case-insensitive;
      * online: the same test is case-SENSITIVE and appends "ctc" to the type;
      * offline: an unknown or empty model_type routes to the transducer operator (:50-52) -- the graph comes from the architecture keys;
      * online: no default case -- an unknown type is refused with the reason."""
    from k2transducerasr_amd.k2w import read_k2w, write_k2w
    from k2transducerasr_amd.synth import write_synthetic_model

    def variant(src, name, **changes):
        meta, tensors = read_k2w(src)
        for k, v in changes.items():
            if v is None:
                meta.pop(k, None)
            else:
                meta[k] = v
        p = str(tmp_path / f"{name}.k2w")
        write_k2w(p, meta, tensors.items())
        return p

    ctc_src = str(tmp_path / "ctc.k2w")
    write_synthetic_model(ctc_src, "zipformer2-ctc-tiny-test")
    ctc_stream_src = str(tmp_path / "ctc_stream.k2w")
    write_synthetic_model(ctc_stream_src, "zipformer2-ctc-streaming-tiny-test")
    mt = lambda p: run(api_driver, "meta", p, "model_type")   # noqa: E731
    # offline CTC export: model_type zipformer2, the comment says ctc
    assert mt(variant(ctc_src, "a", model_type="zipformer2", comment="streaming ctc zipformer2")) == "zipformer2ctc"
    assert mt(variant(ctc_src, "b", model_type="zipformer2", comment="CTC head, Zipformer2 encoder")) == "zipformer2ctc"     # ToLower (:59)
    assert mt(variant(ctc_src, "c", model_type="", comment="zipformer2 + ctc")) == "zipformer2ctc"
    # the comment does not say both words: the transducer type stands
    assert mt(variant(tiny_model_path, "d", model_type="zipformer2", comment="zipformer2 transducer")) == "zipformer2"
    assert mt(variant(tiny_model_path, "e", model_type="zipformer2", comment="ctc only")) == "zipformer2"
    # unknown / empty / absent model_type: OfflineRecognizer.cs:50-52's default operator; the graph from the architecture keys
    for i, given in enumerate(["", "zipformer2_v9", None]):
        p = variant(tiny_model_path, f"f{i}", model_type=given, comment="")
        assert mt(p) == "zipformer2"
        if given:
            assert run(api_driver, "meta", p, "model_type_as_given") == given
    conf = str(tmp_path / "conf.k2w")
    write_synthetic_model(conf, "conformer-tiny-test")
    assert mt(variant(conf, "g", model_type="", comment="")) == "conformer"
    # online: case-sensitive (OnlineModel.cs:103), model_type + "ctc"
    assert mt(variant(ctc_stream_src, "h", model_type="zipformer2", comment="streaming ctc zipformer2")) == "zipformer2ctc"
    assert mt(variant(streaming_tiny_path, "i", model_type="zipformer2", comment="Streaming CTC Zipformer2")) == "zipformer2"     # no ToLower online
    assert mt(variant(ctc_stream_src, "j", model_type="zipformer2ctc", comment="")) == "zipformer2ctc"
    # online: OnlineRecognizer.cs:26-44 has no default case
    out = mt(variant(streaming_tiny_path, "k", model_type="", comment=""))
    assert out.startswith("ERR -6") and "no default case" in out


def test_large_vocabulary_repack_under_sanitizers(api_driver, tmp_path):
    """csrc/model.cpp packs an f16 copy of joiner.output_linear in MFMA fragment order and computes the per-column error bounds for
    vocabularies >= 1024 (the greedy search's screening pass): index arithmetic over (tile, K step, lane, element) and a hand-written
    float -> half conversion -- run here under ASan + UBSan through the real loader, on a vocabulary that is no multiple of 16 and with
    weights that overflow f16, are subnormal in f16, and are not finite."""
    import numpy as np
    from k2transducerasr_amd.k2w import read_k2w, write_k2w
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path / "wide.k2w")
    write_synthetic_model(p, "zipformer2-tiny-test", meta_overrides={"vocab_size": "1101"})
    assert run(api_driver, "meta", p, "vocab_size") == "1101"
    meta, tensors = read_k2w(p)
    w = tensors["joiner.output_linear.weight"].copy()
    w[3, 0], w[4, 1], w[5, 2], w[6, 3], w[7, 4] = 1e6, -7e4, 3e-8, np.inf, np.nan
    tensors["joiner.output_linear.weight"] = w
    p2 = str(tmp_path / "wide_odd.k2w")
    write_k2w(p2, meta, tensors.items())
    assert run(api_driver, "meta", p2, "model_type") == "zipformer2"


def test_host_layer_from_several_threads_under_thread_sanitizer(streaming_tiny_path, tiny_model_path):
    """INTEGRATION.md's threading contract -- calls on one handle serialise on its mutex, different handles run concurrently, any
    thread may call -- checked by ThreadSanitizer (`make tsan`: csrc/api.cpp + model.cpp + tunables.cpp over the CPU stand-in of the
    engine, whose counters and free lists are as unsynchronised as the real engine's): six threads with OnlineStreams of their own on
    ONE shared streaming handle, three with OfflineStreams on ONE shared offline handle, two creating and destroying handles of their
    own meanwhile; the first two handles of the process are opened from two threads at the same moment (that raced on the switches'
    read-the-environment-once flag before it became a std::call_once).  Any report fails."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "k2transducerasr_amd", "csrc"), "-s", "tsan"])
    exe = os.path.join(ROOT, "tests", "native", "k2hip_tsan_api_driver")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([exe, streaming_tiny_path, tiny_model_path, "6", "1500"], capture_output=True, text=True, env=env, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert r.stdout.startswith("threads ok:")
    steps = int(r.stdout.split(":")[1].split()[0])
    assert steps > 500
