// IOnlineProj on the MI355X engine: the operator OnlineRecognizer.InitHip (csharp/OnlineRecognizer.Hip.cs) selects for
// decodingMethod "greedy_search_operators", when the UNCHANGED ForwardBatchGreedySearch loop (OnlineRecognizer.cs:85-219) is to
// run against GPU operators.  (The faster route replaces the loop itself: OnlineRecognizer.Hip.cs.)  Lives inside the K2TransducerAsr assembly because the operator methods of
// IOnlineProj are `internal` (IOnlineProj.cs:65-71).  Source only (no dotnet toolchain in the build image); the same call sequence
// is exercised through ctypes by tests/test_online_gpu.py::test_operator_level_online_proj_runs_the_reference_loop.
//
// States: the reference moves every stream's caches through managed arrays each tick (GetEncoderInitStates -> stack_states ->
// EncoderProj -> unstack_states, OnlineProjOfZipformer2.cs:144-618).  Here a stream's caches are one slot of the device state pool
// (k2hip_online_state_t) that EncoderProj advances in place.  What travels through the managed lists is a two-float TOKEN array
// per stream; a ConditionalWeakTable ties the native handle's lifetime to it, so when the OnlineStream (and with it the token)
// is collected the slot is returned to the pool.
using System;
using System.Collections.Generic;
using System.Linq;
using System.Runtime.CompilerServices;
using K2TransducerAsr.Hip;
using K2TransducerAsr.Model;
using Microsoft.ML.OnnxRuntime;

namespace K2TransducerAsr
{
    internal sealed class HipOnlineState
    {
        internal IntPtr Handle;
        internal HipOnlineState(IntPtr h) { Handle = h; }
        ~HipOnlineState() { if (Handle != IntPtr.Zero) { K2Hip.k2hip_online_state_destroy(Handle); Handle = IntPtr.Zero; } }
    }

    internal class OnlineProjOfHip : IOnlineProj, IDisposable
    {
        private IntPtr _model;
        private readonly bool _ownsModel;
        private K2HipModelInfo _info;
        private OnlineCustomMetadata _customMetadata = new OnlineCustomMetadata();
        private readonly ConditionalWeakTable<float[], HipOnlineState> _states = new ConditionalWeakTable<float[], HipOnlineState>();
        private int _chunkLength, _shiftLength, _framesPerChunk;

        public OnlineProjOfHip(string k2wPath, int device = 0)
        {
            K2Hip.Check(K2Hip.k2hip_model_create(k2wPath, null, device, out _model), "OnlineProjOfHip: model load failed");
            _ownsModel = true;
            Init();
        }

        // over a model the recognizer already holds (OnlineRecognizer.InitHip); the recognizer disposes of it
        internal OnlineProjOfHip(HipOnlineModel model)
        {
            _model = model.Handle;
            _ownsModel = false;
            Init();
        }

        private void Init()
        {
            K2Hip.Check(K2Hip.k2hip_model_get_info(_model, out _info), "OnlineProjOfHip: model info failed");
            K2Hip.Check(K2Hip.k2hip_online_chunk_info(_model, out _chunkLength, out _shiftLength, out _framesPerChunk),
                        "OnlineProjOfHip: not a streaming model");
            _customMetadata.Context_size = _info.context_size;   // OnlineModel.cs reads the same keys from the .onnx metadata
            _customMetadata.Vocab_size = _info.vocab_size;
            _customMetadata.Joiner_dim = _info.joiner_dim;
            _customMetadata.Model_type = K2Hip.Meta(_model, "model_type") ?? "zipformer2";
            _customMetadata.Version = K2Hip.Meta(_model, "version");
            _customMetadata.Model_author = K2Hip.Meta(_model, "model_author");
            _customMetadata.Comment = K2Hip.Meta(_model, "comment");
        }

        public InferenceSession EncoderSession { get => null; set { } }
        public InferenceSession DecoderSession { get => null; set { } }
        public InferenceSession JoinerSession { get => null; set { } }
        public OnlineCustomMetadata CustomMetadata { get => _customMetadata; set => _customMetadata = value; }
        public int Blank_id { get; set; } = 0;
        public int Sos_eos_id { get; set; } = 1;
        public int Unk_id { get; set; } = 2;
        public int ChunkLength { get => _chunkLength; set { } }   // T of the export (OnlineProjOfZipformer2.cs:80-92)
        public int ShiftLength { get => _shiftLength; set { } }
        public int FeatureDim { get => _info.feature_dim; set { } }
        public int SampleRate { get; set; } = 16000;

        // :144-238 -- a fresh (zeroed) slot; the shape List<List<float[]>> is kept, its one leaf is the token
        public List<List<float[]>> GetEncoderInitStates(int batchSize = 1)
        {
            K2Hip.Check(K2Hip.k2hip_online_state_create(_model, out IntPtr h), "GetEncoderInitStates failed");
            var token = new float[2];
            _states.Add(token, new HipOnlineState(h));
            return new List<List<float[]>> { new List<float[]> { token } };
        }

        // :240-340 -- the identity on handles: one inner list holding the B tokens in batch order
        public List<List<float[]>> stack_states(List<List<List<float[]>>> stateList)
        {
            return new List<List<float[]>> { stateList.Select(s => s[0][0]).ToList() };
        }

        // :342-489
        public List<List<List<float[]>>> unstack_states(List<float[]> encoder_out_states)
        {
            return encoder_out_states.Select(t => new List<List<float[]>> { new List<float[]> { t } }).ToList();
        }

        // :491-618.  modelInputs[i].Speech is the stream's GetDecodeChunk (ChunkLength x FeatureDim raw fbank frames).
        public EncoderOutputEntity EncoderProj(List<OnlineInputEntity> modelInputs, int batchSize, List<List<float[]>> statesList)
        {
            int chunkFloats = _chunkLength * _info.feature_dim;
            var x = new float[(long)batchSize * chunkFloats];
            var handles = new IntPtr[batchSize];
            List<float[]> tokens = statesList[0];
            for (int i = 0; i < batchSize; i++)
            {
                Array.Copy(modelInputs[i].Speech, 0, x, (long)i * chunkFloats, chunkFloats);
                if (!_states.TryGetValue(tokens[i], out HipOnlineState st)) throw new Exception("EncoderProj: unknown state");
                handles[i] = st.Handle;
            }
            var o = new EncoderOutputEntity();
            o.encoder_out = new float[(long)batchSize * _framesPerChunk * _info.joiner_dim];
            try
            {
                K2Hip.Check(K2Hip.k2hip_online_encoder(_model, handles, batchSize, x, o.encoder_out, o.encoder_out.LongLength), "EncoderProj failed");
            }
            catch (Exception ex) { throw new Exception("EncoderProj failed", ex); }
            o.encoder_out_lens = Enumerable.Repeat((long)_framesPerChunk, batchSize).ToArray();
            o.encoder_out_states = tokens;   // advanced in place
            GC.KeepAlive(statesList);
            return o;
        }

        public DecoderOutputEntity DecoderProj(Int64[] decoder_input, int batchSize)
        {
            var o = new DecoderOutputEntity();
            int n = decoder_input == null ? batchSize : decoder_input.Length / _info.context_size;
            o.decoder_out = new float[(long)n * _info.joiner_dim];
            K2Hip.Check(K2Hip.k2hip_decoder(_model, decoder_input, n, o.decoder_out), "DecoderProj failed");
            return o;
        }

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
            if (_model != IntPtr.Zero && _ownsModel) K2Hip.k2hip_model_destroy(_model);
            _model = IntPtr.Zero;
        }
        void IOnlineProj.Dispose() => Dispose();
    }
}
