// What a maintainer adds to K2TransducerAsr/OfflineRecognizer.cs to route the hot loop to the GPU.
// (1) Proj selection, next to OfflineRecognizer.cs:38-53:
//         case "zipformer2" when encoderFilePath.EndsWith(".k2w"):
//             _offlineProj = new OfflineProjOfHip(encoderFilePath);
//             decodingMethod = "greedy_search_hip";
//             break;
// (2) delegate selection, next to :54-68:
//         case "greedy_search_hip":
//             _forward = new ForwardOffline(this.ForwardGreedySearchHip);
//             _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchHip);
//             break;
// (3) the two delegates: the whole body of ForwardBatchGreedySearch (:189-303) becomes one native call.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using K2TransducerAsr.Hip;

namespace K2TransducerAsr
{
    public partial class OfflineRecognizer
    {
        private void ForwardBatchGreedySearchHip(List<OfflineStream> streams)
        {
            var proj = (OfflineProjOfHip)_offlineProj;
            int B = streams.Count, maxLen = 0;
            var pins = new GCHandle[B];
            var ptrs = new IntPtr[B];
            var lens = new long[B];
            try
            {
                for (int i = 0; i < B; i++)
                {
                    float[] f = streams[i].OfflineInputEntity.Speech;
                    pins[i] = GCHandle.Alloc(f, GCHandleType.Pinned);
                    ptrs[i] = pins[i].AddrOfPinnedObject();
                    lens[i] = streams[i].OfflineInputEntity.SpeechLength;
                    maxLen = Math.Max(maxLen, (int)lens[i]);
                }
                int T = (maxLen + 80 * 19) / proj.FeatureDim;                       // PadHelper.cs:17,22
                int maxTokens = Math.Max(1, K2Hip.k2hip_encoder_out_frames(proj.Handle, T));
                var tok = new long[(long)B * maxTokens];
                var ts = new int[(long)B * maxTokens];
                var n = new int[B];
                K2Hip.Check(K2Hip.k2hip_offline_greedy(proj.Handle, ptrs, lens, B, tok, ts, n, maxTokens),
                            "Offline recognition failed");                            // same message as :299-302
                for (int m = 0; m < B; m++)
                {
                    var tokens = new List<Int64>();
                    var stamps = new List<int>();
                    for (int i = 0; i < B; i++) { tokens.Add(_blank_id); tokens.Add(_blank_id); stamps.Add(0); stamps.Add(0); } // :250-267
                    for (int k = 0; k < n[m]; k++) { tokens.Add(tok[(long)m * maxTokens + k]); stamps.Add(ts[(long)m * maxTokens + k]); }
                    streams[m].Tokens = tokens;                                      // :292
                    streams[m].Timestamps.AddRange(stamps);                          // :293
                    streams[m].RemoveSamples();                                      // :294
                }
            }
            finally
            {
                foreach (var p in pins) if (p.IsAllocated) p.Free();
            }
        }

        private void ForwardGreedySearchHip(OfflineStream stream)
        {
            var proj = (OfflineProjOfHip)_offlineProj;
            float[] f = stream.OfflineInputEntity.Speech;
            int T = (stream.OfflineInputEntity.SpeechLength + 80 * 19) / proj.FeatureDim;
            int maxTokens = Math.Max(1, K2Hip.k2hip_encoder_out_frames(proj.Handle, T));
            var tok = new long[maxTokens];
            var ts = new int[maxTokens];
            var n = new int[1];
            K2Hip.Check(K2Hip.k2hip_offline_greedy_single(proj.Handle, f, stream.OfflineInputEntity.SpeechLength, tok, ts, n, maxTokens),
                        "Offline recognition failed");                                // :183-186
            var hyp = new List<Int64> { -1, _blank_id };                             // :115-117
            for (int k = 0; k < n[0]; k++) { hyp.Add(tok[k]); stream.Timestamps.Add(ts[k]); }
            stream.Tokens = hyp;                                                     // :180
        }
    }
}
