// The other halves of `partial class OnlineRecognizer` and `partial class OnlineStream`: what routes the STREAMING hot path
// to libk2hip.so.  Source only (no dotnet toolchain in the build image).
//
// How it plugs into the reference tree (csharp/patches/OnlineRecognizer.cs.patch and OnlineStream.cs.patch are the exact
// edits; `patch -p1` applies them):
//   * OnlineRecognizer.cs:11  `public class OnlineRecognizer` -> `public partial class OnlineRecognizer`
//   * OnlineRecognizer.cs:21  IN FRONT of `OnlineModel onlineModel = new OnlineModel(encoderFilePath, ...)`:
//         if (Hip.K2Hip.IsK2w(encoderFilePath)) { InitHip(encoderFilePath, decoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim); return; }
//     (`new OnlineModel` opens ONNXRuntime sessions on the path, OnlineModel.cs:28-30,224-228, and leaves `CustomMetadata` null
//     without an encoder session, :32 -- the switch at :26 would dereference it.)  `onlineModel` is a local of the constructor
//     (:21-44): nothing behind the constructor uses it, so there is no other use site to serve.
//   * OnlineRecognizer.cs:62  CreateOnlineStream: `if (_hipModel != null && _hipFused) return new OnlineStream(_hipModel);`
//   * OnlineRecognizer.cs:499 Dispose: `_hipModel?.Dispose();` BEHIND the closing brace of `if (_onlineProj != null) { _onlineProj.Dispose(); }` --
//     after the operator (OnlineProjOfHip borrows the handle) but not inside that block: on the fused route `_onlineProj` is null and the
//     block is skipped, and the model (weights, arenas, the streams' state pool) must be released all the same
//   * WHICH GPU: the constructor keeps its signature; the device rides on the paths (K2Hip.SplitSpec): encoderFilePath "model.k2w@3", or
//     decoderFilePath "device=3" -- unused otherwise on this route.  Streams are pinned to the recognizer's GPU at CreateOnlineStream time
//     (OnlineRecognizer.cs:60-64); a host spreads its streams over N recognizers, stream id mod N (SURVEY 8e, INTEGRATION.md)
//   * OnlineStream.cs:7       `public class OnlineStream` -> `public partial class OnlineStream`
//   * OnlineStream.cs:59,126  AddSamples / IsFinished forward to the native stream when it exists; :164 Dispose destroys it
//
// The reference's IOnlineProj contract (IOnlineProj.cs:65-71) moves every stream's caches through managed arrays each tick
// (GetEncoderInitStates / stack_states / EncoderProj / unstack_states).  On the GPU the caches never leave HBM, so the unit of the
// fast route is the STREAM: an OnlineStream that owns a native handle, and one delegate that replaces ForwardBatchGreedySearch
// (OnlineRecognizer.cs:85-219) by one k2hip_online_step per tick.
//
// decodingMethod on a .k2w model:
//   "greedy_search" (default)    the fused route above (any streaming model type of the container, CTC included: k2hip_online_step
//                                runs the model's own search, as :33-36 would select it)
//   "greedy_search_operators"    the reference's UNCHANGED loop (:85-219) over OnlineProjOfHip's operators (csharp/OnlineProjOfHip.cs;
//                                streaming Zipformer2 transducers only), for A/B comparisons against the ONNX path
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using K2TransducerAsr.Hip;

namespace K2TransducerAsr
{
    internal sealed class HipOnlineModel : IDisposable
    {
        internal IntPtr Handle;
        internal int ChunkLength, ShiftLength, FramesPerChunk;

        internal HipOnlineModel(string k2wPath, int device = 0)
        {
            // (OnlineModel.cs:101-106's `comment` -> "...ctc" rule and the decoder / joiner metadata are applied by the library when it
            // reads the container: csrc/model.cpp)
            K2Hip.Check(K2Hip.k2hip_model_create(k2wPath, null, device, out Handle), "OnlineRecognizer: model load failed");
            K2Hip.Check(K2Hip.k2hip_online_chunk_info(Handle, out ChunkLength, out ShiftLength, out FramesPerChunk),
                        "OnlineRecognizer: not a streaming model");
        }
        public void Dispose() { if (Handle != IntPtr.Zero) { K2Hip.k2hip_model_destroy(Handle); Handle = IntPtr.Zero; } }
    }

    // the members OnlineStream gains (partial class): the native stream replaces _states, _wavFrontend and the
    // feature FIFO; Hyp / Tokens / Timestamps keep their managed types and are refreshed after every step.
    public partial class OnlineStream
    {
        internal IntPtr HipStream = IntPtr.Zero;

        internal OnlineStream(HipOnlineModel model)
        {
            K2Hip.Check(K2Hip.k2hip_online_stream_create(model.Handle, out HipStream), "OnlineStream: create failed");
            _hyp = new Int64[] { 0, 0 };                     // OnlineStream.cs:43-45
            _tokens = new List<Int64> { 0, 0 };
        }

        // AddSamples (:57-79), IsFinished (:124-161) forward when HipStream != IntPtr.Zero:
        internal void AddSamplesHip(float[] samples) =>
            K2Hip.Check(K2Hip.k2hip_online_stream_accept_samples(HipStream, samples, samples.LongLength), "AddSamples failed");

        internal bool IsFinishedHip(bool isEndpoint)
        {
            K2Hip.Check(K2Hip.k2hip_online_stream_is_finished(HipStream, isEndpoint ? 1 : 0, out int fin), "IsFinished failed");
            return fin != 0;
        }

        // Dispose(bool) (:162-190) calls this first
        internal void DisposeHip()
        {
            if (HipStream != IntPtr.Zero) { K2Hip.k2hip_online_stream_destroy(HipStream); HipStream = IntPtr.Zero; }
        }

        internal void PullResultsHip()
        {
            int n = K2Hip.k2hip_online_stream_num_tokens(HipStream);
            var tok = new long[n];
            K2Hip.Check(K2Hip.k2hip_online_stream_get_tokens(HipStream, tok, n), "get_tokens failed");
            int m = K2Hip.k2hip_online_stream_num_timestamps(HipStream);
            var ts = new int[m];
            K2Hip.Check(K2Hip.k2hip_online_stream_get_timestamps(HipStream, ts, m), "get_timestamps failed");
            _tokens = tok.ToList();                          // OnlineRecognizer.cs:209
            _timestamps = ts.ToList();                       // :210 (the native list already holds the AddRange result)
            K2Hip.Check(K2Hip.k2hip_online_stream_get_hyp(HipStream, _hyp), "get_hyp failed");   // :208
        }
    }

    public partial class OnlineRecognizer
    {
        private HipOnlineModel _hipModel;
        private bool _hipFused;

        // the constructor's early branch (see the header): everything :21-57 does, for a .k2w container
        private void InitHip(string encoderFilePath, string decoderFilePath, string tokensFilePath, string decodingMethod, int sampleRate, int featureDim)
        {
            K2Hip.SplitSpec(encoderFilePath, decoderFilePath, out string k2wPath, out int device);
            _hipModel = new HipOnlineModel(k2wPath, device);
            _tokens = File.ReadAllLines(tokensFilePath);                                // :24
            _hipFused = decodingMethod != "greedy_search_operators";
            if (_hipFused)
            {
                _forwardBatch = new ForwardBatchOnline(this.ForwardBatchGreedySearchHip);
                // _onlineProj stays null: the loops that dereference it (:91-204, :226-305) are not bound, CreateOnlineStream returns
                // the native-handle stream, Dispose (:496) checks for null -- and releases _hipModel behind that check, not inside it
            }
            else
            {
                var proj = new OnlineProjOfHip(_hipModel);                              // borrows the handle; :26-44's selection
                proj.SampleRate = sampleRate;                                           // :23
                _onlineProj = proj;
                _forwardBatch = new ForwardBatchOnline(this.ForwardBatchGreedySearch);  // the reference's own loop, :48-50
            }
        }

        // replaces ForwardBatchGreedySearch (:85-219): GetDecodeChunk, stack_states, EncoderProj, the 8-frame
        // greedy loop, unstack_states and RemoveChunk are one native call; streams without a full chunk are
        // removed from the caller's list exactly as :117-120 does.
        private void ForwardBatchGreedySearchHip(List<OnlineStream> streams)
        {
            if (streams.Count == 0) return;
            try
            {
                int B = streams.Count;
                var handles = streams.Select(s => s.HipStream).ToArray();
                var decoded = new int[B];
                var nNew = new int[B];
                K2Hip.Check(K2Hip.k2hip_online_step(_hipModel.Handle, handles, B, decoded, nNew), "Online recognition failed");
                var skipped = new List<OnlineStream>();
                for (int i = 0; i < B; i++)
                {
                    if (decoded[i] == 0) skipped.Add(streams[i]);
                    else streams[i].PullResultsHip();
                }
                foreach (var s in skipped) streams.Remove(s);
            }
            catch (Exception ex)
            {
                throw new Exception("Online recognition failed", ex);       // same outer message as :214-217
            }
        }
    }
}
