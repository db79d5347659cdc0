#!/usr/bin/env python3
"""Regenerates csharp/patches/*.patch: the edits a maintainer makes INSIDE the reference's own files so that the classes under
csharp/ plug in (unified diffs against /root/reference, two lines of context; the new code itself lives in csharp/*.Hip.cs as the
other halves of the `partial` classes, so the diffs stay a few lines each).  tests/test_csharp_patch.py applies them to a copy
of the reference tree and checks what they promise (the early branch sits in front of `new OfflineModel(...)`, every later use of
the model object is served, the classes are partial).  Run here only: the reference tree does not exist on the GPU box."""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("K2_REFERENCE", "/root/reference")

# (file, [(anchor line (stripped, must be unique), "replace" | "before" | "after", [new lines])])
EDITS = {
    "K2TransducerAsr/OfflineRecognizer.cs": [
        ("public class OfflineRecognizer : IDisposable", "replace", ["    public partial class OfflineRecognizer : IDisposable"]),
        ("_offlineModel = new OfflineModel(encoderFilePath, decoderFilePath, joinerFilePath, threadsNum);", "before", [
            "            // MI355X engine (csharp/OfflineRecognizer.Hip.cs): a .k2w container instead of the three ONNX files.  This branch must sit",
            "            // HERE, in front of `new OfflineModel(...)`: that constructor opens ONNXRuntime sessions on the paths it is given.",
            "            if (Hip.K2Hip.IsK2w(encoderFilePath))",
            "            {",
            "                InitHip(encoderFilePath, decoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim);",
            "                return;",
            "            }",
        ]),
        ("OfflineStream offlineStream = new OfflineStream(_offlineModel.CustomMetadata, sampleRate: _frontendConfEntity.fs, featureDim: _frontendConfEntity.n_mels);", "before", [
            "            // the fused route: the stream owns a native handle, AddSamples queues raw samples, the fbank runs on the GPU inside GetResults",
            "            if (_hipSamples) return new OfflineStream((OfflineProjOfHip)_offlineProj, _offlineModel.CustomMetadata);   // OfflineRecognizer.Hip.cs",
        ]),
    ],
    "K2TransducerAsr/OfflineStream.cs": [
        ("public class OfflineStream", "replace", ["    public partial class OfflineStream"]),
        ("public void AddSamples(float[] samples)", "after+1", [
            "            // in FRONT of the static lock below: on this route no CPU fbank runs and nothing shared is touched (OfflineRecognizer.Hip.cs)",
            "            if (HipStream != IntPtr.Zero) { AddSamplesHip(samples); return; }",
        ]),
        ("if (_wavFrontend != null)", "before", [
            "                DisposeHip();   // OfflineRecognizer.Hip.cs: k2hip_offline_stream_destroy",
        ]),
    ],
    "K2TransducerAsr/OnlineRecognizer.cs": [
        ("public class OnlineRecognizer", "replace", ["    public partial class OnlineRecognizer"]),
        ("OnlineModel onlineModel = new OnlineModel(encoderFilePath, decoderFilePath, joinerFilePath, configFilePath: configFilePath, threadsNum: threadsNum);", "before", [
            "            // MI355X engine (csharp/OnlineRecognizer.Hip.cs): must precede `new OnlineModel(...)`, which opens ONNXRuntime sessions and",
            "            // leaves CustomMetadata null when there is no encoder session (OnlineModel.cs:32).",
            "            if (Hip.K2Hip.IsK2w(encoderFilePath))",
            "            {",
            "                InitHip(encoderFilePath, decoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim);",
            "                return;",
            "            }",
        ]),
        ("OnlineStream onlineStream = new OnlineStream(_onlineProj);", "before", [
            "            if (_hipModel != null && _hipFused) return new OnlineStream(_hipModel);   // the stream owns a native handle (OnlineRecognizer.Hip.cs)",
        ]),
        # BEHIND the closing brace of `if (_onlineProj != null) { ... }`: on the fused route _onlineProj is null and that block is skipped
        ("_onlineProj.Dispose();", "after-block", [
            "                _hipModel?.Dispose();   // after the operator (OnlineProjOfHip borrows this handle), outside its null check",
        ]),
    ],
    "K2TransducerAsr/OnlineStream.cs": [
        ("public class OnlineStream", "replace", ["    public partial class OnlineStream"]),
        ("public void AddSamples(float[] samples)", "after+1", [
            "            if (HipStream != IntPtr.Zero) { AddSamplesHip(samples); return; }   // OnlineRecognizer.Hip.cs",
        ]),
        ("public bool IsFinished(bool isEndpoint = false)", "after+1", [
            "            if (HipStream != IntPtr.Zero) return IsFinishedHip(isEndpoint);     // OnlineRecognizer.Hip.cs",
        ]),
        ("if (_wavFrontend != null)", "before", [
            "                DisposeHip();   // OnlineRecognizer.Hip.cs: k2hip_online_stream_destroy",
        ]),
    ],
}


def patched(lines, edits, name):
    out = list(lines)
    for anchor, how, new in edits:
        hits = [i for i, ln in enumerate(out) if ln.strip() == anchor]
        assert len(hits) == 1, f"{name}: anchor {anchor!r} found {len(hits)} times"
        i = hits[0]
        new = [n + "\n" for n in new]
        if how == "replace":
            out[i: i + 1] = new
        elif how == "before":
            out[i:i] = new
        elif how == "after":
            out[i + 1: i + 1] = new
        elif how == "after-block":  # behind the closing brace that follows the anchor
            assert out[i + 1].strip() == "}", (name, anchor)
            out[i + 2: i + 2] = new
        elif how == "after+1":      # behind the opening brace that follows the anchor
            assert out[i + 1].strip() == "{", (name, anchor)
            out[i + 2: i + 2] = new
        else:
            raise ValueError(how)
    return out


def main():
    for rel, edits in EDITS.items():
        src = open(os.path.join(REF, rel), encoding="utf-8-sig", newline="").read()
        crlf = "\r\n" in src
        lines = src.replace("\r\n", "\n").splitlines(keepends=True)
        new = patched(lines, edits, rel)
        diff = list(difflib.unified_diff(lines, new, "a/" + rel, "b/" + rel, n=2))
        out = os.path.join(ROOT, "csharp", "patches", os.path.basename(rel) + ".patch")
        with open(out, "w", encoding="utf-8") as f:
            f.writelines(diff)
        print(f"{out}: {sum(1 for d in diff if d.startswith('+') and not d.startswith('+++'))} lines added, "
              f"{sum(1 for d in diff if d.startswith('-') and not d.startswith('---'))} removed{' (reference file has CRLF line ends)' if crlf else ''}")


if __name__ == "__main__":
    sys.exit(main())
