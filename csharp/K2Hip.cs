// P/Invoke declarations for libk2hip.so (include/k2hip.h).
// Source only: there is no dotnet toolchain in the build image, so this file is not compiled here.
using System;
using System.Runtime.InteropServices;

namespace K2TransducerAsr.Hip
{
    [StructLayout(LayoutKind.Sequential)]
    internal struct K2HipModelInfo
    {
        public int vocab_size, context_size, joiner_dim, feature_dim, sample_rate, num_stacks, device, reserved;
    }

    internal static class K2Hip
    {
        private const string Lib = "k2hip";

        [DllImport(Lib)] internal static extern IntPtr k2hip_last_error();
        [DllImport(Lib)] internal static extern int k2hip_model_create(string weightsPath, string overrides, int device, out IntPtr model);
        [DllImport(Lib)] internal static extern int k2hip_model_destroy(IntPtr model);
        [DllImport(Lib)] internal static extern int k2hip_device_count();
        [DllImport(Lib)] internal static extern int k2hip_model_get_info(IntPtr model, out K2HipModelInfo info);
        [DllImport(Lib)] internal static extern long k2hip_fbank_num_frames(IntPtr model, long nSamples);
        [DllImport(Lib)] internal static extern int k2hip_fbank(IntPtr model, float[] samples, long n, float[] feats, long capFrames, out long nFrames);
        [DllImport(Lib)] internal static extern int k2hip_encoder_out_frames(IntPtr model, int T);
        [DllImport(Lib)] internal static extern int k2hip_offline_encoder(IntPtr model, float[] x, long[] xLens, int B, int T,
            float[] encOut, long capFloats, long[] encOutLens, out int Tprime);
        [DllImport(Lib)] internal static extern int k2hip_decoder(IntPtr model, long[] y, int N, float[] decOut);
        [DllImport(Lib)] internal static extern int k2hip_joiner(IntPtr model, float[] enc, float[] dec, int N, float[] logits);
        [DllImport(Lib)] internal static extern int k2hip_offline_greedy(IntPtr model, IntPtr[] feats, long[] nFloats, int B,
            long[] tokens, int[] timestamps, int[] nTokens, int maxTokens);
        [DllImport(Lib)] internal static extern int k2hip_offline_greedy_single(IntPtr model, float[] feats, long nFloats,
            long[] tokens, int[] timestamps, int[] nTokens, int maxTokens);

        // ---- OfflineStream / OfflineRecognizer.GetResults on native streams (include/k2hip.h, "OfflineStream" section): AddSamples queues
        // raw samples, GetResults runs fbank + pad + encoder + search for the whole batch on the device
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_create(IntPtr model, out IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_destroy(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_accept_samples(IntPtr stream, float[] samples, long n);
        [DllImport(Lib)] internal static extern long k2hip_offline_stream_speech_length(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_recognizer_get_results(IntPtr model, IntPtr[] streams, int B);
        [DllImport(Lib)] internal static extern int k2hip_offline_recognizer_get_result(IntPtr model, IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_num_tokens(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_num_timestamps(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_get_tokens(IntPtr stream, long[] tokens, int cap);
        [DllImport(Lib)] internal static extern int k2hip_offline_stream_get_timestamps(IntPtr stream, int[] timestamps, int cap);

        // ---- streaming path (include/k2hip.h, "streaming path" section)
        [DllImport(Lib)] internal static extern int k2hip_online_chunk_info(IntPtr model, out int chunkLength, out int shiftLength, out int framesPerChunk);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_create(IntPtr model, out IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_destroy(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_reset(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_accept_samples(IntPtr stream, float[] samples, long n);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_accept_features(IntPtr stream, float[] feats, long nFrames);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_is_finished(IntPtr stream, int isEndpoint, out int finished);
        [DllImport(Lib)] internal static extern int k2hip_online_step(IntPtr model, IntPtr[] streams, int B, int[] decoded, int[] nNewTokens);
        [DllImport(Lib)] internal static extern int k2hip_online_state_create(IntPtr model, out IntPtr state);
        [DllImport(Lib)] internal static extern int k2hip_online_state_destroy(IntPtr state);
        [DllImport(Lib)] internal static extern long k2hip_online_state_processed_len(IntPtr state);
        [DllImport(Lib)] internal static extern int k2hip_online_encoder(IntPtr model, IntPtr[] states, int B, float[] feats, float[] encoderOut, long capFloats);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_num_tokens(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_num_timestamps(IntPtr stream);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_get_tokens(IntPtr stream, long[] tokens, int cap);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_get_timestamps(IntPtr stream, int[] timestamps, int cap);
        [DllImport(Lib)] internal static extern int k2hip_online_stream_get_hyp(IntPtr stream, long[] hyp2);

        [DllImport(Lib)] internal static extern int k2hip_model_meta(IntPtr model, string key, byte[] buf, int cap);
        [DllImport(Lib)] internal static extern int k2hip_set_decoding_method(IntPtr model, string method, int beam);

        // Which GPU?  The reference's constructors (OfflineRecognizer.cs:27-28, OnlineRecognizer.cs:18-19) take file paths and nothing
        // else, and they stay as they are: the device rides on the paths.
        //   encoderFilePath "model.k2w@3"                 -> container "model.k2w" on device 3 (the suffix is split off only when what
        //                                                    follows the LAST '@' is 1 - 4 decimal digits: k2hip_parse_model_spec's rule)
        //   decoderFilePath "device=3" (encoder "model.k2w") -> the same; the decoder path is otherwise unused on this route (a .k2w
        //                                                    container holds all three networks)
        // An explicit "@N" wins over "device=N"; neither: device 0.  A host that serves N GPUs builds N recognizers, one per device, from
        // N threads (INTEGRATION.md "More than one GPU"; tests/native/multi_handle_host.c is that host in C).
        internal static void SplitSpec(string encoderFilePath, string decoderFilePath, out string path, out int device)
        {
            path = encoderFilePath;
            device = 0;
            bool fromSpec = false;
            if (!string.IsNullOrEmpty(encoderFilePath))
            {
                int at = encoderFilePath.LastIndexOf('@');
                int nd = encoderFilePath.Length - at - 1;
                if (at > 0 && nd >= 1 && nd <= 4)
                {
                    int d = 0;
                    bool digits = true;
                    for (int i = at + 1; i < encoderFilePath.Length; i++)
                    {
                        char ch = encoderFilePath[i];
                        if (ch < '0' || ch > '9') { digits = false; break; }
                        d = d * 10 + (ch - '0');
                    }
                    if (digits) { path = encoderFilePath.Substring(0, at); device = d; fromSpec = true; }
                }
            }
            if (!fromSpec && !string.IsNullOrEmpty(decoderFilePath) && decoderFilePath.StartsWith("device=", StringComparison.Ordinal))
            {
                if (int.TryParse(decoderFilePath.Substring(7), out int d) && d >= 0) device = d;
            }
        }

        // Is this "encoder file" a .k2w weights container (the engine's format) rather than an ONNX file?  Decided by the file's
        // magic, not by its name, so a renamed container still routes here and an ONNX file never does.  (k2w.py: the file starts
        // with the ASCII bytes "K2W1".)  A "@N" device suffix (SplitSpec) is not part of the file name.
        internal static bool IsK2w(string encoderFilePath)
        {
            SplitSpec(encoderFilePath, null, out string path, out _);
            if (string.IsNullOrEmpty(path) || !System.IO.File.Exists(path)) return false;
            try
            {
                using (var f = System.IO.File.OpenRead(path))
                {
                    var magic = new byte[4];
                    return f.Read(magic, 0, 4) == 4 && magic[0] == (byte)'K' && magic[1] == (byte)'2' && magic[2] == (byte)'W' && magic[3] == (byte)'1';
                }
            }
            catch (System.IO.IOException) { return false; }
        }

        // CustomMetadataMap[key] of the weights container; null when the key is absent (k2hip_model_meta -> K2HIP_ERR_INVALID)
        internal static string Meta(IntPtr model, string key)
        {
            var buf = new byte[4096];
            if (k2hip_model_meta(model, key, buf, buf.Length) != 0) return null;
            int n = Array.IndexOf(buf, (byte)0);
            return System.Text.Encoding.UTF8.GetString(buf, 0, n < 0 ? buf.Length : n);
        }

        internal static void Check(int rc, string what)
        {
            if (rc != 0)
                throw new Exception(what, new Exception(Marshal.PtrToStringAnsi(k2hip_last_error())));
        }
    }
}
