"""The reference-side binding must be CONSTRUCTIBLE: csharp/patches/*.patch are the edits inside the reference's own files, the
classes under csharp/ are the other halves of the `partial` classes.  There is no dotnet toolchain here, so nothing is compiled;
what can be checked is checked: the patches apply cleanly to the reference tree as it is, the early branch ends up in front of the
constructors that would open ONNXRuntime sessions on a .k2w path, every later use of the model object is one the shim serves, and
every P/Invoke the shims declare is an export of include/k2hip.h with the same number of parameters.
(The reference tree exists in the build container only; on the GPU box the tests that need it skip.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ["OfflineRecognizer.cs", "OfflineStream.cs", "OnlineRecognizer.cs", "OnlineStream.cs"]

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "K2TransducerAsr")) or not shutil.which("patch"),
                               reason="the reference tree / patch(1) is not here")


@pytest.fixture(scope="module")
def patched_tree(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref")
    os.makedirs(d / "K2TransducerAsr")
    for f in FILES:
        shutil.copy(os.path.join(REF, "K2TransducerAsr", f), d / "K2TransducerAsr" / f)
    for f in FILES:
        p = os.path.join(ROOT, "csharp", "patches", f + ".patch")
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-d", str(d), "-i", p], capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, (f, r.stdout, r.stderr)
    return d


def _src(tree, f):
    return open(tree / "K2TransducerAsr" / f, encoding="utf-8-sig").read()


@needs_ref
def test_patches_apply_and_put_the_branch_in_front_of_the_onnx_constructors(patched_tree):
    off = _src(patched_tree, "OfflineRecognizer.cs")
    on = _src(patched_tree, "OnlineRecognizer.cs")
    st = _src(patched_tree, "OnlineStream.cs")
    assert "public partial class OfflineRecognizer" in off and "public partial class OnlineRecognizer" in on and "public partial class OnlineStream" in st
    # the branch is the first statement of the constructor body, in front of the session-opening constructor
    assert 0 < off.index("K2Hip.IsK2w(encoderFilePath)") < off.index("_offlineModel = new OfflineModel(encoderFilePath")
    assert 0 < on.index("K2Hip.IsK2w(encoderFilePath)") < on.index("onlineModel = new OnlineModel(encoderFilePath")
    ctor = off[off.index("public OfflineRecognizer("):]
    body = ctor[ctor.index("{") + 1:]
    first_stmt = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith("//")][0]
    assert first_stmt.startswith("if (Hip.K2Hip.IsK2w("), first_stmt
    assert "return;" in body[: body.index("_offlineModel = new OfflineModel(")]
    # streams of the fused route own a native handle; AddSamples / IsFinished / Dispose reach it
    assert on.index("new OnlineStream(_hipModel)") < on.index("new OnlineStream(_onlineProj)")
    for call in ("AddSamplesHip(samples)", "IsFinishedHip(isEndpoint)", "DisposeHip()"):
        assert call in st
    # OfflineStream (round 5): on the fused route AddSamples hands the RAW samples to the native stream -- the branch sits in front of
    # `lock (obj)` and of `_wavFrontend.GetFbank(samples)` (OfflineStream.cs:45-47), so no CPU fbank runs and no static lock is taken
    os_ = _src(patched_tree, "OfflineStream.cs")
    assert "public partial class OfflineStream" in os_
    add = os_[os_.index("public void AddSamples(float[] samples)"):]
    assert 0 < add.index("AddSamplesHip(samples); return;") < add.index("lock (obj)") < add.index("_wavFrontend.GetFbank(samples)")
    assert "DisposeHip();" in os_[os_.index("protected virtual void Dispose(bool disposing)"):]
    create = off[off.index("public OfflineStream CreateOfflineStream()"):]
    assert 0 < create.index("if (_hipSamples) return new OfflineStream((OfflineProjOfHip)_offlineProj") < create.index("new OfflineStream(_offlineModel.CustomMetadata, sampleRate")
    # OnlineRecognizer.Dispose (round 4's advice): the model handle is released BEHIND the `if (_onlineProj != null) { ... }` block, not
    # inside it -- on the fused route _onlineProj is null and a Dispose inside the block would never run (the GPU memory of the model leaked)
    disp = on[on.index("protected virtual void Dispose(bool disposing)"):]
    m = re.search(r"if \(_onlineProj != null\)\s*\{\s*_onlineProj\.Dispose\(\);\s*\}\s*_hipModel\?\.Dispose\(\);", disp)
    assert m, "_hipModel?.Dispose() must follow the closing brace of the _onlineProj null check"
    # regenerating the patches from tools/make_csharp_patches.py gives what is committed
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "make_csharp_patches.py")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert subprocess.run(["git", "diff", "--quiet", "--", "csharp/patches"], cwd=ROOT).returncode == 0, "csharp/patches is stale"


@needs_ref
def test_every_use_of_the_model_objects_is_served_by_the_shim():
    """OfflineRecognizer keeps `_offlineModel` as a field: every line that touches it must be either inside the constructor part the
    early branch skips, or a read of `.CustomMetadata` (which InitHip assigns) -- and csharp/OfflineRecognizer.Hip.cs must NAME each
    such line.  `onlineModel` is a constructor local of OnlineRecognizer: no use behind the constructor."""
    ref = open(os.path.join(REF, "K2TransducerAsr", "OfflineRecognizer.cs"), encoding="utf-8-sig").read().splitlines()
    hip = open(os.path.join(ROOT, "csharp", "OfflineRecognizer.Hip.cs")).read()
    header = hip[: hip.index("using System;")]
    ctor_end = next(i for i, ln in enumerate(ref) if "public OfflineStream CreateOfflineStream()" in ln)
    later = [(i + 1, ln.strip()) for i, ln in enumerate(ref) if "_offlineModel" in ln and i >= ctor_end]
    assert later, "the reference no longer uses _offlineModel behind its constructor?"
    for no, ln in later:
        assert "_offlineModel.CustomMetadata" in ln, f"OfflineRecognizer.cs:{no} uses _offlineModel for something InitHip does not provide: {ln}"
        assert re.search(rf":{no}\b", header), f"csharp/OfflineRecognizer.Hip.cs does not name the use site OfflineRecognizer.cs:{no}"
    assert "_offlineModel.CustomMetadata = proj.CustomMetadata" in hip and 'new OfflineModel("", "", ""' in hip
    # initModel must really return null for an empty path, or `new OfflineModel("", "", "")` would open a session
    om = open(os.path.join(REF, "K2TransducerAsr", "OfflineModel.cs"), encoding="utf-8-sig").read()
    assert re.search(r"if \(string\.IsNullOrEmpty\(modelFilePath\) \|\| !File\.Exists\(modelFilePath\)\)\s*\{\s*return null;", om)
    assert re.search(r"public OfflineCustomMetadata CustomMetadata \{ get => _customMetadata; set => _customMetadata = value; \}", om)
    on = open(os.path.join(REF, "K2TransducerAsr", "OnlineRecognizer.cs"), encoding="utf-8-sig").read().splitlines()
    ctor_end = next(i for i, ln in enumerate(on) if "public OnlineStream CreateOnlineStream()" in ln)
    assert not [ln for ln in on[ctor_end:] if re.search(r"\bonlineModel\b", ln)]
    # the fields the partial halves assign exist under these names in the reference
    joined = "\n".join(ref)
    for field in ("_offlineModel", "_offlineProj", "_tokens", "_frontendConfEntity", "_wavFrontend", "_forward", "_forwardBatch", "_blank_id"):
        assert re.search(rf"private [\w\?\[\]<>]+ {field}\b", joined), field
    joined_on = "\n".join(on)
    for field in ("_tokens", "_onlineProj", "_forwardBatch"):
        assert re.search(rf"private [\w\?\[\]<>]+ {field}\b", joined_on), field


def test_pinvoke_declarations_match_the_header():
    """every [DllImport] in csharp/K2Hip.cs names an export declared in include/k2hip.h, with as many parameters"""
    hdr = open(os.path.join(ROOT, "include", "k2hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(k2hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    cs = open(os.path.join(ROOT, "csharp", "K2Hip.cs")).read()
    seen = 0
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*internal static extern \w+ (k2hip_\w+)\(([^;]*?)\);", cs, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if not args else len(args.split(","))
        assert name in decl, f"K2Hip.cs imports {name}, which include/k2hip.h does not declare"
        assert decl[name] == n, f"{name}: {n} parameters in K2Hip.cs, {decl[name]} in include/k2hip.h"
        seen += 1
    assert seen >= 30
    # and everything the shims call through K2Hip exists there
    for f in os.listdir(os.path.join(ROOT, "csharp")):
        if f.endswith(".cs") and f != "K2Hip.cs":
            for name in set(re.findall(r"K2Hip\.(k2hip_\w+)\(", open(os.path.join(ROOT, "csharp", f)).read())):
                assert re.search(rf"extern \w+ {name}\(", cs), f"{f} calls K2Hip.{name}, which K2Hip.cs does not declare"


def test_the_fused_offline_route_never_touches_the_cpu_front_end():
    """csharp/OfflineRecognizer.Hip.cs: the members OfflineStream gains on the fused route must not name `_wavFrontend`, the forward
    delegates must call the native GetResults / GetResult (fbank on the GPU, from the queued samples), and SpeechLength must come from
    the native stream (= 80 x frames, what OfflineStream.cs:55 would hold)."""
    hip = open(os.path.join(ROOT, "csharp", "OfflineRecognizer.Hip.cs")).read()
    code = hip[hip.index("using System;"):]
    stream_part = code[code.index("public partial class OfflineStream"):code.index("public partial class OfflineRecognizer")]
    assert "_wavFrontend" not in stream_part and "GetFbank" not in stream_part
    assert "k2hip_offline_stream_accept_samples(HipStream, samples" in stream_part
    assert stream_part.count("SpeechLength = (int)K2Hip.k2hip_offline_stream_speech_length(HipStream)") >= 2
    rec_part = code[code.index("public partial class OfflineRecognizer"):]
    assert "k2hip_offline_recognizer_get_results(proj.Handle, handles, B)" in rec_part
    assert "k2hip_offline_recognizer_get_result(proj.Handle, stream.HipStream)" in rec_part
    # the feature entry stays reachable only through the operator route ("greedy_search_operators": managed streams, CPU fbank)
    assert "k2hip_offline_greedy(" not in rec_part
    assert re.search(r'case "greedy_search_operators":[^}]*?_hipSamples = false;', rec_part, flags=re.S)


def test_every_recognizer_constructor_path_can_reach_a_device_other_than_zero():
    """The reference's constructors have no device parameter (OfflineRecognizer.cs:27-28, OnlineRecognizer.cs:18-19) and keep their
    signatures: both InitHip halves take encoderFilePath AND decoderFilePath, resolve (path, device) with K2Hip.SplitSpec and hand the
    device to the native model constructor; the patches pass decoderFilePath through."""
    for f, ctor in (("OfflineRecognizer.Hip.cs", "new OfflineProjOfHip(k2wPath, device)"), ("OnlineRecognizer.Hip.cs", "new HipOnlineModel(k2wPath, device)")):
        src = open(os.path.join(ROOT, "csharp", f)).read()
        body = src[src.index("private void InitHip(string encoderFilePath, string decoderFilePath,"):]
        assert body.index("K2Hip.SplitSpec(encoderFilePath, decoderFilePath, out string k2wPath, out int device);") < body.index(ctor), f
    for f in ("OfflineRecognizer.cs.patch", "OnlineRecognizer.cs.patch"):
        assert "InitHip(encoderFilePath, decoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim);" in open(os.path.join(ROOT, "csharp", "patches", f)).read(), f
    for f, call in (("OfflineProjOfHip.cs", "k2hip_model_create(k2wPath, null, device, out _model)"), ("OnlineRecognizer.Hip.cs", "k2hip_model_create(k2wPath, null, device, out Handle)")):
        assert call in open(os.path.join(ROOT, "csharp", f)).read(), f
    # the managed rule is the library's (k2hip_parse_model_spec): the LAST '@', 1 - 4 decimal digits behind it, "device=N" as the fallback
    cs = open(os.path.join(ROOT, "csharp", "K2Hip.cs")).read()
    split = cs[cs.index("internal static void SplitSpec("):cs.index("internal static bool IsK2w(")]
    assert "LastIndexOf('@')" in split and "nd >= 1 && nd <= 4" in split and 'StartsWith("device=", StringComparison.Ordinal)' in split
    assert "SplitSpec(encoderFilePath, null, out string path, out _);" in cs[cs.index("internal static bool IsK2w("):]


def test_model_spec_parser_of_the_library():
    """k2hip_parse_model_spec: "path@N" -> (path, N); anything else is a path on device 0.  Pure string work (runs without a GPU)."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from k2transducerasr_amd import load_library
    L = load_library()
    L.k2hip_parse_model_spec.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
    cases = [("m.k2w", "m.k2w", 0), ("m.k2w@3", "m.k2w", 3), ("/a@b/m.k2w@12", "/a@b/m.k2w", 12), ("m.k2w@", "m.k2w@", 0), ("m.k2w@x1", "m.k2w@x1", 0),
             ("@7", "@7", 0), ("m@12345", "m@12345", 0), ("user@host/m.k2w", "user@host/m.k2w", 0), ("m.k2w@0007", "m.k2w", 7)]
    for spec, path, dev in cases:
        buf, d = C.create_string_buffer(256), C.c_int32(-1)
        assert L.k2hip_parse_model_spec(spec.encode(), buf, 256, C.byref(d)) == 0
        assert (buf.value.decode(), d.value) == (path, dev), spec
    buf, d = C.create_string_buffer(4), C.c_int32(0)
    assert L.k2hip_parse_model_spec(b"model.k2w@1", buf, 4, C.byref(d)) == -5   # K2HIP_ERR_CAPACITY
