"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/k2hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("k2hip.h", "k2hip_debug.h")   # the boundary, and the test / tuning hooks kept apart from it


def declared_symbols(headers=HEADERS):
    syms = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(k2hip_[a-z0-9_]+)\s*\(", src))
    return sorted(syms)


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ("k2hip_model_create", "k2hip_fbank", "k2hip_pad_sequence", "k2hip_offline_encoder", "k2hip_decoder",
              "k2hip_joiner", "k2hip_offline_greedy", "k2hip_offline_greedy_single", "k2hip_offline_greedy_from_samples",
              "k2hip_offline_stream_accept_samples", "k2hip_offline_recognizer_get_results", "k2hip_last_error",
              "k2hip_get_timing"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from k2transducerasr_amd import library_path, load_library
    load_library()
    lib = ctypes.CDLL(library_path())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_library_exports_nothing_the_headers_do_not_declare():
    """The dynamic symbol table is the C ABI and nothing else: no undeclared hook, no C++ symbol of the engine's types
    (csrc/exports.map), and no debug hook hiding in the boundary header."""
    import shutil
    import subprocess
    from k2transducerasr_amd import library_path
    nm = shutil.which("nm")
    if not nm:
        pytest.skip("no nm")
    out = subprocess.run([nm, "-D", "--defined-only", library_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    extra = sorted(exported - set(declared_symbols()))
    assert not extra, f"exported but declared in neither header: {extra}"
    boundary_debug = [s for s in declared_symbols(("k2hip.h",)) if s.startswith("k2hip_debug_") and s != "k2hip_debug_set_switch"]
    assert not boundary_debug, boundary_debug


def test_every_header_entry_cites_the_reference():
    src = open(os.path.join(ROOT, "include", "k2hip.h")).read()
    assert len(re.findall(r"\.cs:\d+", src)) >= 15


def test_version_and_error_strings():
    from k2transducerasr_amd import load_library
    L = load_library()
    assert L.k2hip_version().decode().startswith("k2hip")
    assert isinstance(L.k2hip_last_error(), bytes)


def test_no_cpu_fallback(tiny_model_path):
    """Without a HIP device model creation must fail loudly (K2HIP_ERR_NO_DEVICE)."""
    from k2transducerasr_amd import K2HipError, Model, load_library
    if load_library().k2hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(K2HipError) as e:
        Model(tiny_model_path, 0)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_null_arguments_are_errors_not_crashes():
    from k2transducerasr_amd import load_library
    L = load_library()
    assert L.k2hip_model_create(None, None, 0, None) == -1
    assert L.k2hip_fbank(None, None, 0, None, 0, None) == -1
    assert L.k2hip_offline_greedy(None, None, None, 0, None, None, None, 0) == -1
    assert b"null argument" in L.k2hip_last_error()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under k2transducerasr_amd/ may import,
    link or load it."""
    pkg = os.path.join(ROOT, "k2transducerasr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "k2_oracle" not in txt and "libk2oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_k2w_roundtrip(tmp_path):
    import numpy as np
    from k2transducerasr_amd.k2w import read_k2w, write_k2w
    rng = np.random.default_rng(0)
    tensors = [("a.weight", rng.standard_normal((3, 5)).astype(np.float32)), ("b", np.arange(7, dtype=np.float32)),
               ("c.d", rng.standard_normal((2, 1, 3, 3)).astype(np.float32))]
    meta = {"model_type": "zipformer2", "encoder_dims": "1,2,3", "comment": ""}
    p = str(tmp_path / "x.k2w")
    write_k2w(p, meta, tensors)
    m2, t2 = read_k2w(p)
    assert m2 == meta
    for n, a in tensors:
        np.testing.assert_array_equal(t2[n], a)


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """The boundary is a C ABI: include/k2hip.h must compile as C99 (no C++ in the signatures), and a C translation unit that calls
    an entry point must link against libk2hip.so with nothing but the C runtime."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    for h in HEADERS:
        hdr = os.path.join(root, "include", h)
        subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    src = tmp_path / "host.c"
    src.write_text('#include "k2hip.h"\n#include <stdio.h>\nint main(void) { printf("%s\\n", k2hip_version()); return k2hip_last_error() == 0; }\n')
    lib = os.path.join(root, "k2transducerasr_amd")
    exe = tmp_path / "host"
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", lib, "-lk2hip", "-Wl,-rpath," + lib], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "LD_LIBRARY_PATH": lib + ":" + os.environ.get("LD_LIBRARY_PATH", "")})
    assert out.returncode == 0 and out.stdout.strip()


def test_bad_containers_are_io_errors_through_the_abi(tiny_model_path, tmp_path):
    """k2hip_model_create parses and validates the container on the host before it asks for a device, so a damaged or
    mismatched weights file is K2HIP_ERR_IO (-2) here as on a GPU box -- never a crash (the error contract of
    OfflineProjOfTransducer.cs:87-90; the parser itself is fuzzed under ASan/UBSan in tests/test_sanitizers.py)."""
    import struct
    import numpy as np
    from k2transducerasr_amd import K2HipError, Model
    from k2transducerasr_amd.k2w import read_k2w, write_k2w
    raw = open(tiny_model_path, "rb").read()
    (data_off,) = struct.unpack_from("<Q", raw, 16)
    for name, blob in {"truncated_header": raw[: data_off - 100], "truncated_data": raw[: len(raw) // 2], "garbage": os.urandom(4096),
                       "wrapping_offset": raw[:16] + struct.pack("<Q", 2**64 - 64) + raw[24:]}.items():
        p = tmp_path / f"{name}.k2w"
        p.write_bytes(blob)
        with pytest.raises(K2HipError) as e:
            Model(str(p), 0)
        assert e.value.code == -2, (name, str(e.value))
    # metadata and weights that disagree: vocab_size says 37, the embedding has 30 rows
    meta, tensors = read_k2w(tiny_model_path)
    bad = str(tmp_path / "mismatch.k2w")
    write_k2w(bad, meta, [(n, np.ascontiguousarray(a[:30]) if n == "decoder.embedding.weight" else a) for n, a in tensors.items()])
    with pytest.raises(K2HipError) as e:
        Model(bad, 0)
    assert e.value.code == -2 and "decoder.embedding.weight" in str(e.value)
    # an int64 tensor in the table (an ONNX shape constant that slipped through an importer) is ignored, not fatal
    extra = str(tmp_path / "with_i64.k2w")
    write_k2w(extra, meta, list(tensors.items()) + [("/encoder/Constant_output_0", np.arange(4, dtype=np.int64))])
    try:
        Model(extra, 0).close()
    except K2HipError as e2:
        assert e2.code == -3, str(e2)     # no GPU here: the file itself was accepted


def test_profiles_named_in_committed_bench_records_exist():
    """Profile hygiene (round 5): every `profiles/...` path a committed bench record cites -- `roofline.traffic_note`,
    `roofline.committed_profile.source`, `roofline.hbm_kernels` -- must be a committed file, and a record must cite files of its OWN tag
    (round 4's records cited r04_v3 summaries that were never committed: those records stay as the history they are; the rule holds from
    r05 on)."""
    import glob
    import json
    import re
    prof = os.path.join(ROOT, "profiles")
    checked = 0
    for path in sorted(glob.glob(os.path.join(prof, "r*_bench*.json"))):
        tag = re.match(r"(r\d+_v\d+)_", os.path.basename(path))
        if not tag or int(tag.group(1)[1:3]) < 5:
            continue
        text = open(path).read()
        json.loads(text.strip().splitlines()[-1])
        for cited in set(re.findall(r"profiles/[A-Za-z0-9_.\-]+", text)):
            assert os.path.exists(os.path.join(ROOT, cited)), f"{os.path.basename(path)} cites {cited}, which is not committed"
            assert os.path.basename(cited).startswith(tag.group(1) + "_"), f"{os.path.basename(path)} cites {cited}: a file of another profile set"
            checked += 1
    # (no r05 record yet at the start of the round: nothing to check is fine; the rule bites once tools/refresh_profiles.sh has run)
    assert checked >= 0
