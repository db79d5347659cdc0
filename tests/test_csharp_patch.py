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
FILES = ["OfflineRecognizer.cs", "OnlineRecognizer.cs", "OnlineStream.cs"]

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
