// The other halves of `partial class OfflineRecognizer` and `partial class OfflineStream`: what routes OfflineRecognizer's hot path
// to libk2hip.so.  Source only (no dotnet toolchain in the build image).
//
// How it plugs into the reference tree (csharp/patches/OfflineRecognizer.cs.patch and OfflineStream.cs.patch are the exact edits;
// `patch -p1` applies them):
//   * OfflineRecognizer.cs:12   `public class OfflineRecognizer`  ->  `public partial class OfflineRecognizer`
//   * OfflineRecognizer.cs:30   IN FRONT of `_offlineModel = new OfflineModel(encoderFilePath, ...)`:
//         if (Hip.K2Hip.IsK2w(encoderFilePath)) { InitHip(encoderFilePath, decoderFilePath, tokensFilePath, decodingMethod, sampleRate, featureDim); return; }
//     The branch cannot live in the Proj switch at :38-53: `new OfflineModel` (:30) has by then handed the path to
//     `new InferenceSession(path)` (OfflineModel.cs:25,111-115), which throws on a .k2w, and the switch key
//     `_offlineModel.CustomMetadata.Model_type` only exists once ONNX metadata was read.
//   * OfflineRecognizer.cs:73   CreateOfflineStream: `if (_hipSamples) return new OfflineStream((OfflineProjOfHip)_offlineProj, _offlineModel.CustomMetadata);`
//   * OfflineStream.cs:7        `public class OfflineStream` -> `public partial class OfflineStream`
//   * OfflineStream.cs:45       AddSamples, IN FRONT of `lock (obj)`: `if (HipStream != IntPtr.Zero) { AddSamplesHip(samples); return; }`
//   * OfflineStream.cs:73       Dispose(bool): `DisposeHip();`
// Nothing else changes.  Every later use of `_offlineModel` in OfflineRecognizer.cs reads `CustomMetadata` only:
//   :31  FeatureDim (set again by InitHip)      :73  CreateOfflineStream -> new OfflineStream(_offlineModel.CustomMetadata, ...)
//   :95, :191, :307, :367  `_offlineModel.CustomMetadata.Context_size` at the top of the four Forward* loops
// InitHip serves them by constructing `new OfflineModel("", "", "", n)` -- initModel returns null for an empty path
// (OfflineModel.cs:86-89), so no session is opened and the `!= null` guards at :31-72 skip the metadata reads -- and assigning the
// engine's metadata (filled from k2hip_model_get_info / k2hip_model_meta) to its settable `CustomMetadata` (:78).
// Dispose (:586-611) already calls `_offlineProj.Dispose()`, which is OfflineProjOfHip.Dispose -> k2hip_model_destroy.
//
// WHERE THE FBANK RUNS.  The reference computes it on the CPU inside OfflineStream.AddSamples (`_wavFrontend.GetFbank(samples)`,
// OfflineStream.cs:47) under a process-wide static lock (:16,45): 32 utterances of a batch are 32 serialized CPU fbanks before GetResults
// starts.  On the fused route a stream owns a NATIVE stream (k2hip_offline_stream_*): AddSamples hands the raw samples over (a copy, no
// lock, no `_wavFrontend`), `OfflineInputEntity.SpeechLength` is kept equal to what the reference would show (80 x the frame count of
// the samples so far: k2hip_offline_stream_speech_length), and GetResults is ONE native call for the batch --
// k2hip_offline_recognizer_get_results: samples H2D, one batched fbank launch, pad, encoder, search, tokens D2H -- after which the
// native stream holds Tokens / Timestamps exactly as OfflineRecognizer.cs:250-296 leaves them (2 x B blank prefix, Timestamps.AddRange,
// RemoveSamples) and this file copies them into the managed lists.
//
// WHICH GPU.  The constructors keep their signatures; the device rides on the paths (K2Hip.SplitSpec): encoderFilePath
// "model.k2w@3", or decoderFilePath "device=3" -- unused otherwise on this route.  INTEGRATION.md "More than one GPU".
//
// decodingMethod on a .k2w model:
//   "greedy_search" (default)    the fused delegates below: fbank + pad + encoder + loop are ONE native call per batch
//   "modified_beam_search"       the same entry under k2hip_set_decoding_method(.., beam 4) (BASELINE configs[2]; no reference counterpart)
//   "greedy_search_operators"    the reference's UNCHANGED loops (:93-303) over OfflineProjOfHip's three operators (one native call
//                                per frame) with the reference's own managed streams (CPU fbank) -- for A/B comparisons against the ONNX path
//   a zipformer2ctc container    the fused delegates as well (the native entries run the CTC search for such a model, k2hip.h);
//                                "greedy_search_operators" keeps the reference's unchanged CTC loops (:305-424) over OfflineProjOfHip.EncoderProj
using System;
using System.Collections.Generic;
using System.IO;
using K2TransducerAsr.Hip;
using K2TransducerAsr.Model;

namespace K2TransducerAsr
{
    // the members OfflineStream gains (partial class): on the fused route the native stream replaces `_wavFrontend` and the feature
    // buffer; Tokens / Timestamps keep their managed types and are refreshed after every GetResults / GetResult.
    public partial class OfflineStream
    {
        internal IntPtr HipStream = IntPtr.Zero;

        // what OfflineStream(OfflineCustomMetadata, int, int) (:19-35) does, without the CPU front end
        internal OfflineStream(OfflineProjOfHip proj, OfflineCustomMetadata offlineCustomMetadata)
        {
            _offlineCustomMetadata = offlineCustomMetadata;                    // :21
            _offlineInputEntity = new OfflineInputEntity();                    // :22
            _tokens = new List<Int64> { _blank_id, _blank_id };                // :34
            K2Hip.Check(K2Hip.k2hip_offline_stream_create(proj.Handle, out HipStream), "OfflineStream: create failed");
        }

        // AddSamples (:43-57) forwards here when HipStream != IntPtr.Zero -- in front of `lock (obj)`: nothing shared is touched
        internal void AddSamplesHip(float[] samples)
        {
            K2Hip.Check(K2Hip.k2hip_offline_stream_accept_samples(HipStream, samples, samples.LongLength), "AddSamples failed");
            // Speech itself never comes to the host on this route; its LENGTH is what the reference would hold (:55)
            _offlineInputEntity.SpeechLength = (int)K2Hip.k2hip_offline_stream_speech_length(HipStream);
        }

        // Dispose(bool) (:70-91) calls this first
        internal void DisposeHip()
        {
            if (HipStream != IntPtr.Zero) { K2Hip.k2hip_offline_stream_destroy(HipStream); HipStream = IntPtr.Zero; }
        }

        // after a native GetResults / GetResult: Tokens (:292 / :180), Timestamps (:293 / :181; the native list already holds the
        // AddRange result) and the length RemoveSamples (:294, OfflineStream.cs:58-68) leaves
        internal void PullResultsHip()
        {
            int n = K2Hip.k2hip_offline_stream_num_tokens(HipStream);
            var tok = new long[n];
            K2Hip.Check(K2Hip.k2hip_offline_stream_get_tokens(HipStream, tok, n), "get_tokens failed");
            int m = K2Hip.k2hip_offline_stream_num_timestamps(HipStream);
            var ts = new int[Math.Max(m, 1)];
            K2Hip.Check(K2Hip.k2hip_offline_stream_get_timestamps(HipStream, ts, ts.Length), "get_timestamps failed");
            _tokens = new List<Int64>(tok);
            _timestamps = new List<int>(m);
            for (int i = 0; i < m; i++) _timestamps.Add(ts[i]);
            _offlineInputEntity.SpeechLength = (int)K2Hip.k2hip_offline_stream_speech_length(HipStream);
            if (_offlineInputEntity.SpeechLength == 0) _offlineInputEntity.Speech = null;
        }
    }

    public partial class OfflineRecognizer
    {
        private bool _hipSamples;   // the fused route: streams own native handles, the fbank runs on the GPU inside GetResults

        // the constructor's early branch (see the header): everything :30-68 does, for a .k2w container
        private void InitHip(string encoderFilePath, string decoderFilePath, string tokensFilePath, string decodingMethod, int sampleRate, int featureDim)
        {
            K2Hip.SplitSpec(encoderFilePath, decoderFilePath, out string k2wPath, out int device);
            var proj = new OfflineProjOfHip(k2wPath, device);
            _offlineProj = proj;
            _offlineModel = new OfflineModel("", "", "", 1);          // no sessions: initModel("") returns null (OfflineModel.cs:86-89)
            _offlineModel.CustomMetadata = proj.CustomMetadata;       // what :73, :95, :191, :307, :367 read
            _offlineModel.FeatureDim = featureDim;                    // :31
            _tokens = File.ReadAllLines(tokensFilePath);              // :32
            _frontendConfEntity = new FrontendConfEntity();           // :34-37
            _frontendConfEntity.fs = sampleRate;
            _frontendConfEntity.n_mels = featureDim;
            _wavFrontend = new WavFrontend(_frontendConfEntity);
            bool ctc = proj.CustomMetadata.Model_type == "zipformer2ctc";   // :46-49
            switch (decodingMethod)
            {
                case "greedy_search_operators":                       // managed streams, CPU fbank, the reference's own loops
                    _hipSamples = false;
                    if (ctc)
                    {
                        _forward = new ForwardOffline(this.ForwardGreedySearchCTC);
                        _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchCTC);
                    }
                    else
                    {
                        _forward = new ForwardOffline(this.ForwardGreedySearch);
                        _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearch);
                    }
                    break;
                case "modified_beam_search":
                    K2Hip.Check(K2Hip.k2hip_set_decoding_method(proj.Handle, "modified_beam_search", 4), "OfflineRecognizer: decoding method");
                    _hipSamples = true;
                    _forward = new ForwardOffline(this.ForwardGreedySearchHip);          // (the single-stream path stays greedy, k2hip.h)
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchHip);
                    break;
                default:                                                                 // "greedy_search", "greedy_search_ctc", and :63-66's default
                    _hipSamples = true;
                    _forward = new ForwardOffline(this.ForwardGreedySearchHip);
                    _forwardBatch = new ForwardBatchOffline(this.ForwardBatchGreedySearchHip);
                    break;
            }
        }

        // replaces ForwardBatchGreedySearch (:189-303) -- and, for a CTC container, ForwardBatchGreedySearchCTC (:366-424): one native
        // call for the batch, from the samples the streams were given
        private void ForwardBatchGreedySearchHip(List<OfflineStream> streams)
        {
            var proj = (OfflineProjOfHip)_offlineProj;
            int B = streams.Count;
            var handles = new IntPtr[B];
            for (int i = 0; i < B; i++)
            {
                if (streams[i].HipStream == IntPtr.Zero)
                    throw new Exception("Offline recognition failed", new Exception("stream " + i + " was not created by this recognizer's CreateOfflineStream"));
                handles[i] = streams[i].HipStream;
            }
            K2Hip.Check(K2Hip.k2hip_offline_recognizer_get_results(proj.Handle, handles, B), "Offline recognition failed");   // same message as :299-302
            for (int m = 0; m < B; m++) streams[m].PullResultsHip();                  // :289-296
        }

        // replaces ForwardGreedySearch (:93-187) / ForwardGreedySearchCTC (:305-364)
        private void ForwardGreedySearchHip(OfflineStream stream)
        {
            var proj = (OfflineProjOfHip)_offlineProj;
            if (stream.HipStream == IntPtr.Zero)
                throw new Exception("Offline recognition failed", new Exception("the stream was not created by this recognizer's CreateOfflineStream"));
            K2Hip.Check(K2Hip.k2hip_offline_recognizer_get_result(proj.Handle, stream.HipStream), "Offline recognition failed");   // :183-186
            stream.PullResultsHip();                                                 // :180-181
        }
    }
}
