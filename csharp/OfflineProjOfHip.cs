// IOfflineProj on the MI355X engine.  Lives INSIDE the K2TransducerAsr assembly because the
// operator methods of IOfflineProj are `internal` (IOfflineProj.cs:43-47).
// Source only (no dotnet toolchain in the build image).
using System;
using System.Collections.Generic;
using System.Linq;
using K2TransducerAsr.Hip;
using K2TransducerAsr.Model;
using K2TransducerAsr.Utils;
using Microsoft.ML.OnnxRuntime;

namespace K2TransducerAsr
{
    internal class OfflineProjOfHip : IOfflineProj, IDisposable
    {
        private IntPtr _model;
        private K2HipModelInfo _info;
        private OfflineCustomMetadata _customMetadata = new OfflineCustomMetadata();

        public OfflineProjOfHip(string k2wPath, int device = 0)
        {
            K2Hip.Check(K2Hip.k2hip_model_create(k2wPath, null, device, out _model), "OfflineProjOfHip: model load failed");
            K2Hip.Check(K2Hip.k2hip_model_get_info(_model, out _info), "OfflineProjOfHip: model info failed");
            _customMetadata.Context_size = _info.context_size;   // OfflineModel.cs:33-38
            _customMetadata.Vocab_size = _info.vocab_size;
            _customMetadata.Joiner_dim = _info.joiner_dim;       // OfflineModel.cs:43-45
            // OfflineModel.cs:49-71 reads model_type / version / model_author / comment from the encoder's metadata map; the .k2w
            // container carries the same keys (conformer, lstm, zipformer, zipformer2ctc models route through this operator too)
            _customMetadata.Model_type = K2Hip.Meta(_model, "model_type") ?? "zipformer2";
            _customMetadata.Version = K2Hip.Meta(_model, "version");
            _customMetadata.Model_author = K2Hip.Meta(_model, "model_author");
            _customMetadata.Comment = K2Hip.Meta(_model, "comment");
        }

        internal IntPtr Handle => _model;

        // no ONNXRuntime sessions behind this operator
        public InferenceSession EncoderSession { get => null; set { } }
        public InferenceSession DecoderSession { get => null; set { } }
        public InferenceSession JoinerSession { get => null; set { } }
        public OfflineCustomMetadata CustomMetadata { get => _customMetadata; set => _customMetadata = value; }
        public int Blank_id { get; set; } = 0;     // OfflineModel.cs:18-20
        public int Sos_eos_id { get; set; } = 1;
        public int Unk_id { get; set; } = 2;
        public int FeatureDim => _info.feature_dim;

        // OfflineProjOfTransducer.EncoderProj :48-92 (PadSequence stays managed here; the fused path pads on the GPU)
        public EncoderOutputEntity EncoderProj(List<OfflineInputEntity> modelInputs, int batchSize)
        {
            float[] pad = PadHelper.PadSequence(modelInputs);
            int T = pad.Length / FeatureDim / batchSize;
            int tp = K2Hip.k2hip_encoder_out_frames(_model, T);
            var o = new EncoderOutputEntity();
            o.encoder_out = new float[(long)batchSize * tp * _info.joiner_dim];
            o.encoder_out_lens = new long[batchSize];
            long[] xl = Enumerable.Repeat((long)T, batchSize).ToArray();
            try
            {
                K2Hip.Check(K2Hip.k2hip_offline_encoder(_model, pad, xl, batchSize, T, o.encoder_out, o.encoder_out.LongLength,
                                                        o.encoder_out_lens, out _), "EncoderProj failed");
            }
            catch (Exception ex) { throw new Exception("EncoderProj failed", ex); }   // same outer message as :87-90
            return o;
        }

        // :93-123 (null input -> [-1, blank] per row)
        public DecoderOutputEntity DecoderProj(Int64[] decoder_input, int batchSize)
        {
            var o = new DecoderOutputEntity();
            int n = decoder_input == null ? batchSize : decoder_input.Length / _info.context_size;
            o.decoder_out = new float[(long)n * _info.joiner_dim];
            K2Hip.Check(K2Hip.k2hip_decoder(_model, decoder_input, n, o.decoder_out), "DecoderProj failed");
            return o;
        }

        // :125-152
        public JoinerOutputEntity JoinerProj(float[] encoder_out, float[] decoder_out)
        {
            int n = encoder_out.Length / _info.joiner_dim;
            var o = new JoinerOutputEntity();
            o.Logit = new float[(long)n * _info.vocab_size];
            K2Hip.Check(K2Hip.k2hip_joiner(_model, encoder_out, decoder_out, n, o.Logit), "JoinerProj failed");
            o.Logits = new Microsoft.ML.OnnxRuntime.Tensors.DenseTensor<float>(o.Logit, new[] { n, _info.vocab_size });
            return o;
        }

        public void Dispose()
        {
            if (_model != IntPtr.Zero) { K2Hip.k2hip_model_destroy(_model); _model = IntPtr.Zero; }
        }
    }
}
