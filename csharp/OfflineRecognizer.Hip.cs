// The other half of `partial class OfflineRecognizer`: what routes OfflineRecognizer's hot path to libk2hip.so.
// Source only (no dotnet toolchain in the build image).
//
// How it plugs into the reference tree (csharp/patches/OfflineRecognizer.cs.patch is the exact edit; `patch -p1` applies it):
//   * OfflineRecognizer.cs:12   `public class OfflineRecognizer`  ->  `public partial class OfflineRecognizer`
//   * OfflineRecognizer.cs:30   IN FRONT of `_offlineModel = new OfflineModel(encoderFilePath, ...)`:
//         if (Hip.K2Hip.IsK2w(encoderFilePath)) { InitHip(encoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim); return; }
//     The branch cannot live in the Proj switch at :38-53: `new OfflineModel` (:30) has by then handed the path to
//     `new InferenceSession(path)` (OfflineModel.cs:25,111-115), which throws on a .k2w, and the switch key
//     `_offlineModel.CustomMetadata.Model_type` only exists once ONNX metadata was read.
// Nothing else in OfflineRecognizer.cs changes.  Every later use of `_offlineModel` reads `CustomMetadata` only:
//   :31  FeatureDim (set again by InitHip)      :73  CreateOfflineStream -> new OfflineStream(_offlineModel.CustomMetadata, ...)
//   :95, :191, :307, :367  `_offlineModel.CustomMetadata.Context_size` at the top of the four Forward* loops
// InitHip serves them by constructing `new OfflineModel("", "", "", n)` -- initModel returns null for an empty path
// (OfflineModel.cs:86-89), so no session is opened and the `!= null` guards at :31-72 skip the metadata reads -- and assigning the
// engine's metadata (filled from k2hip_model_get_info / k2hip_model_meta) to its settable `CustomMetadata` (:78).
// Dispose (:586-611) already calls `_offlineProj.Dispose()`, which is OfflineProjOfHip.Dispose -> k2hip_model_destroy.
//
// decodingMethod on a .k2w model:
//   "greedy_search" (default)    the fused delegates below: pad + encoder + loop are ONE native call per batch
//   "modified_beam_search"       the same entry under k2hip_set_decoding_method(.., beam 4) (BASELINE configs[2]; no reference counterpart)
//   "greedy_search_operators"    the reference's UNCHANGED loops (:93-303) over OfflineProjOfHip's three operators (one native call
//                                per frame) -- for A/B comparisons against the ONNX path
//   a zipformer2ctc container    the reference's unchanged CTC loops (:305-424) over OfflineProjOfHip.EncoderProj (log_probs), as
//                                OfflineRecognizer.cs:46-49 selects them
using System;
using System.Collections.Generic;
using System.IO;
using System.Runtime.InteropServices;
using K2TransducerAsr.Hip;
using K2TransducerAsr.Model;

namespace K2TransducerAsr
{
    public partial class OfflineRecognizer
    {
        // the constructor's early branch (see the header): everything :30-68 does, for a .k2w container
        private void InitHip(string k2wPath, string tokensFilePath, string decodingMethod, int sampleRate, int featureDim)
        {
            var proj = new OfflineProjOfHip(k2wPath);
            _offlineProj = proj;
            _offlineModel = new OfflineModel("", "", "", 1);          // no sessions: initModel("") returns null (OfflineModel.cs:86-89)
            _offlineModel.CustomMetadata = proj.CustomMetadata;       // what :73, :95, :191, :307, :367 read
            _offlineModel.FeatureDim = featureDim;                    // :31
            _tokens = File.ReadAllLines(tokensFilePath);              // :32
            _frontendConfEntity = new FrontendConfEntity();           // :34-37
            _frontendConfEntity.fs = sampleRate;
            _frontendConfEntity.n_mels = featureDim;
            _wavFrontend = new WavFrontend(_frontendConfEntity);
            if (proj.CustomMetadata.Model_type == "zipformer2ctc") decodingMethod = "greedy_search_ctc";   // :46-49
            switch (decodingMethod)
            {
                case "greedy_search_ctc":
                    _forward = new ForwardOffline(this.ForwardGreedySearchCTC);
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchCTC);
                    break;
                case "greedy_search_operators":
                    _forward = new ForwardOffline(this.ForwardGreedySearch);
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearch);
                    break;
                case "modified_beam_search":
                    K2Hip.Check(K2Hip.k2hip_set_decoding_method(proj.Handle, "modified_beam_search", 4), "OfflineRecognizer: decoding method");
                    _forward = new ForwardOffline(this.ForwardGreedySearchHip);          // (the single-stream path stays greedy, k2hip.h)
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchHip);
                    break;
                default:                                                                 // "greedy_search", and :63-66's default
                    _forward = new ForwardOffline(this.ForwardGreedySearchHip);
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchHip);
                    break;
            }
        }

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
