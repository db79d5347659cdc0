// What a maintainer adds to K2TransducerAsr to route the STREAMING hot path to the GPU.
// Source only (no dotnet toolchain in the build image).
//
// The reference's IOnlineProj contract (IOnlineProj.cs:65-71) moves every stream's caches through
// managed arrays each tick (GetEncoderInitStates / stack_states / EncoderProj / unstack_states).
// On the GPU the caches never leave HBM, so the drop-in unit is the STREAM: an OnlineStream that owns
// a native handle, and one delegate that replaces ForwardBatchGreedySearch (OnlineRecognizer.cs:85-219).
//
// (1) Proj selection, next to OnlineRecognizer.cs:26-44:
//         case "zipformer2" when encoderFilePath.EndsWith(".k2w"):
//             _hipModel = new HipOnlineModel(encoderFilePath);      // below
//             decodingMethod = "greedy_search_hip";
//             break;
// (2) delegate selection, next to :46-57:
//         case "greedy_search_hip":
//             _forwardBatch = new ForwardBatchOnline(this.ForwardBatchGreedySearchHip);
//             break;
// (3) CreateOnlineStream (:60-64) returns `new OnlineStream(_hipModel)` when _hipModel != null.
using System;
using System.Collections.Generic;
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
