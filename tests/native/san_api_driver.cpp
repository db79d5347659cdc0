// AddressSanitizer / UBSan driver for the host layer of libk2hip (csrc/api.cpp) over a CPU stand-in of the engine
// (engine_stub.cpp).  TEST INFRASTRUCTURE (`make -C k2transducerasr_amd/csrc san`, tests/test_sanitizers.py).
//   san_api_driver online  <streaming.k2w> <seed> <rounds>   random AddSamples / AddFeatures / GetResults / IsFinished / Reset / destroy
//                                                            over several OnlineStreams, every call checked against a plain model
//                                                            of OnlineStream.cs (FIFO length, decodable or not, token bookkeeping)
//   san_api_driver offline <offline.k2w> <seed> <rounds>     OfflineStream AddSamples in pieces, GetResults / GetResult, pipelined
//                                                            submit / wait, operator-level calls with tight output buffers
//   san_api_driver errors  <streaming.k2w>                   null arguments, wrong model, duplicates, poisoned streams
// Any sanitizer report, crash or model mismatch fails the process; prints one summary line.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/k2hip.h"

extern "C" void k2hip_stub_fail_next_step(int n);
extern "C" void k2hip_stub_fail_next_gather_finish(int n);
extern "C" int32_t k2hip_debug_poison_stream(k2hip_online_stream_t* s);
extern "C" int32_t k2hip_debug_stream_mirrored(const k2hip_online_stream_t* s, int32_t* ok);

#define CHECK(cond)                                                                            \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            fprintf(stderr, "CHECK failed at %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, k2hip_last_error()); \
            exit(3);                                                                           \
        }                                                                                      \
    } while (0)
#define OK(call) CHECK((call) == K2HIP_OK)

namespace {

struct RefStream {        // OnlineStream.cs:57-161 as far as the host layer owns it
    long long frames = 0;     // frames in the FIFO (materialised or pending)
    long long rem = 0;        // samples of the incomplete frame shift
    long long pend = 0;       // samples accepted since the last materialisation
    int ntok = 2;
};

long long frames_of(long long n) { return n < 400 ? 0 : 1 + (n - 400) / 160; }

int online(const char* path, unsigned long long seed, int rounds) {
    k2hip_model_t* m = nullptr;
    OK(k2hip_model_create(path, nullptr, 0, &m));
    int32_t T = 0, S = 0, Tp = 0;
    OK(k2hip_online_chunk_info(m, &T, &S, &Tp));
    CHECK(T > S && S > 0 && Tp > 0);
    std::mt19937_64 rng(seed);
    const int N = 7;
    std::vector<k2hip_online_stream_t*> hs(N, nullptr);
    std::vector<RefStream> ref(N);
    for (int i = 0; i < N; i++) OK(k2hip_online_stream_create(m, &hs[i]));
    long long steps = 0, toks = 0;
    std::vector<float> buf;
    for (int r = 0; r < rounds; r++) {
        const int op = (int)(rng() % 10), i = (int)(rng() % N);
        if (op < 4) {          // AddSamples: anything from nothing to 1.5 s, incl. lengths that leave a ragged remainder
            const long long n = rng() % 5 == 0 ? (long long)(rng() % 40) : (long long)(rng() % 24000);
            buf.assign((size_t)n, 0.f);
            for (auto& v : buf) v = (float)((int)(rng() % 2001) - 1000) * 1e-3f;
            OK(k2hip_online_stream_accept_samples(hs[i], n ? buf.data() : nullptr, n));
            // the reference appends whole frames of [remainder ; new]; what is left over waits for the next call
            RefStream& q = ref[i];
            const long long have = q.rem + q.pend + n;
            (void)have;
            q.pend += n;
        } else if (op == 4) {  // AddFeatures directly (materialises pending samples first)
            const long long nf = (long long)(rng() % 90);
            buf.assign((size_t)nf * 80, 1.5f);
            OK(k2hip_online_stream_accept_features(hs[i], nf ? buf.data() : nullptr, nf));
            RefStream& q = ref[i];
            const long long tot = q.rem + q.pend, f = frames_of(tot);
            q.frames += f;
            q.rem = f ? tot - f * 160 : tot;
            q.pend = 0;
            q.frames += nf;
        } else if (op < 8) {   // GetResults over a random subset
            std::vector<k2hip_online_stream_t*> sub;
            std::vector<int> idx;
            for (int k = 0; k < N; k++)
                if (rng() % 3) { sub.push_back(hs[k]); idx.push_back(k); }
            if (sub.empty()) continue;
            std::vector<int32_t> dec(sub.size(), -1), nn(sub.size(), -1);
            OK(k2hip_online_step(m, sub.data(), (int32_t)sub.size(), dec.data(), nn.data()));
            for (size_t k = 0; k < sub.size(); k++) {
                RefStream& q = ref[idx[k]];
                const long long logical = q.frames + frames_of(q.rem + q.pend);
                const int want = logical >= T ? 1 : 0;
                CHECK(dec[k] == want);
                if (want) {   // the pending samples were materialised, a chunk decoded, ShiftLength frames removed
                    const long long tot = q.rem + q.pend, f = frames_of(tot);
                    q.frames += f;
                    q.rem = f ? tot - f * 160 : tot;
                    q.pend = 0;
                    q.frames -= S;
                    CHECK(nn[k] >= 0 && nn[k] <= Tp);
                    q.ntok += nn[k];
                    toks += nn[k];
                    steps++;
                } else {
                    CHECK(nn[k] == 0);
                }
            }
        } else if (op == 8) {  // IsFinished (with its side effect) / token reads
            int32_t fin = -1;
            const int endp = (int)(rng() % 2);
            RefStream& q = ref[i];
            const long long before = q.frames + frames_of(q.rem + q.pend);
            OK(k2hip_online_stream_is_finished(hs[i], endp, &fin));
            CHECK(fin == 0 || fin == 1);
            if (endp) {   // materialises; may append 400 zero samples when at most a chunk is buffered and the frames are not constant
                const long long tot = q.rem + q.pend, f = frames_of(tot);
                q.frames += f;
                q.rem = f ? tot - f * 160 : tot;
                q.pend = 0;
                const long long fl = k2hip_online_stream_speech_length(hs[i]);
                CHECK(fl == (q.frames + frames_of(q.rem)) * 80 || fl == (q.frames + frames_of(q.rem + 400)) * 80);
                if (fl != (q.frames + frames_of(q.rem)) * 80) q.pend = 400;
                CHECK(before <= q.frames + frames_of(q.rem + q.pend));
            }
            const int nt = k2hip_online_stream_num_tokens(hs[i]);
            CHECK(nt == q.ntok);
            std::vector<int64_t> tk((size_t)nt);
            std::vector<int32_t> tsv((size_t)std::max(1, k2hip_online_stream_num_timestamps(hs[i])));
            OK(k2hip_online_stream_get_tokens(hs[i], tk.data(), nt));
            CHECK(k2hip_online_stream_get_tokens(hs[i], tk.data(), nt - 1) == K2HIP_ERR_CAPACITY);
            OK(k2hip_online_stream_get_timestamps(hs[i], tsv.data(), (int32_t)tsv.size()));
            int64_t hyp[2];
            OK(k2hip_online_stream_get_hyp(hs[i], hyp));
            CHECK(hyp[0] == tk[(size_t)nt - 2] && hyp[1] == tk[(size_t)nt - 1]);
        } else {               // Reset, or destroy + create
            if (rng() % 2) {
                OK(k2hip_online_stream_reset(hs[i]));
            } else {
                OK(k2hip_online_stream_destroy(hs[i]));
                OK(k2hip_online_stream_create(m, &hs[i]));
            }
            ref[i] = RefStream();
        }
        const long long fl = k2hip_online_stream_speech_length(hs[i]);
        CHECK(fl == (ref[i].frames + frames_of(ref[i].rem + ref[i].pend)) * 80);
    }
    for (auto* h : hs) OK(k2hip_online_stream_destroy(h));
    OK(k2hip_model_destroy(m));
    printf("online ok: %d rounds, %lld chunk steps, %lld tokens\n", rounds, steps, toks);
    return 0;
}

int offline(const char* path, unsigned long long seed, int rounds) {
    k2hip_model_t* m = nullptr;
    OK(k2hip_model_create(path, nullptr, 0, &m));
    k2hip_model_info info;
    OK(k2hip_model_get_info(m, &info));
    std::mt19937_64 rng(seed);
    long long toks = 0;
    for (int r = 0; r < rounds; r++) {
        const int B = 1 + (int)(rng() % 5);
        std::vector<k2hip_offline_stream_t*> ss(B, nullptr);
        for (int b = 0; b < B; b++) {
            OK(k2hip_offline_stream_create(m, &ss[b]));
            long long total = 0;
            const int pieces = 1 + (int)(rng() % 3);
            for (int p = 0; p < pieces; p++) {
                // (one piece in eight is long enough to outgrow the stream's pinned queue -- 64 K samples at first, pooled buffers larger:
                // the queue is then re-allocated with its contents)
                const long long n = 400 + (long long)(rng() % (rng() % 8 == 0 ? 150000 : 20000));
                std::vector<float> w((size_t)n, 0.25f);
                OK(k2hip_offline_stream_accept_samples(ss[b], w.data(), n));
                total += n;
            }
            // SpeechLength counts the queued samples' frames at once (OfflineStream.cs:55: it is set by AddSamples) ...
            const long long want_len = (long long)k2hip_fbank_num_frames(m, total) * info.feature_dim;
            CHECK(k2hip_offline_stream_speech_length(ss[b]) == want_len && want_len > 0);
            // ... and reading Speech (every other stream: the others keep their samples queued, so that a batch is decoded from
            // samples, from features, or -- mixed -- from features materialised at GetResults) changes nothing visible
            if (rng() % 2) {
                std::vector<float> sp((size_t)want_len);
                OK(k2hip_offline_stream_get_speech(ss[b], sp.data(), (int64_t)sp.size()));
                CHECK(k2hip_offline_stream_get_speech(ss[b], sp.data(), (int64_t)sp.size() - 1) == K2HIP_ERR_CAPACITY);
                CHECK(k2hip_offline_stream_speech_length(ss[b]) == want_len);
            }
        }
        const bool single = B == 1 && rng() % 2;
        if (single) OK(k2hip_offline_recognizer_get_result(m, ss[0]));
        else OK(k2hip_offline_recognizer_get_results(m, ss.data(), B));
        {   // the same stream twice in one batch is refused
            std::vector<k2hip_offline_stream_t*> twice{ss[0], ss[0]};
            CHECK(k2hip_offline_recognizer_get_results(m, twice.data(), 2) == K2HIP_ERR_INVALID);
        }
        for (int b = 0; b < B; b++) {
            const int nt = k2hip_offline_stream_num_tokens(ss[b]);
            CHECK(nt >= 2);
            // RemoveSamples (OfflineStream.cs:58-68): the batch path clears Speech when Tokens.Count > Context_size; the single path keeps it
            const long long after = k2hip_offline_stream_speech_length(ss[b]);
            if (single) CHECK(after > 0);
            else if (nt > info.context_size) CHECK(after == 0);
            std::vector<int64_t> tk((size_t)nt);
            OK(k2hip_offline_stream_get_tokens(ss[b], tk.data(), nt));
            const int nts = k2hip_offline_stream_num_timestamps(ss[b]);
            std::vector<int32_t> tsv((size_t)std::max(nts, 1));
            OK(k2hip_offline_stream_get_timestamps(ss[b], tsv.data(), (int32_t)tsv.size()));
            toks += nts;
            OK(k2hip_offline_stream_destroy(ss[b]));
        }
        // pipelined entries: tickets, output buffers exactly as large as the contract says
        const int Bp = 1 + (int)(rng() % 4);
        const long long n_each = 1600 + (long long)(rng() % 8000);
        std::vector<float> smp((size_t)Bp * n_each, 0.5f);
        const int mt = std::max(1, k2hip_encoder_out_frames(m, (int32_t)k2hip_fbank_num_frames(m, n_each) + 19));
        int32_t t0 = -1, t1 = -1, t2 = -1;
        OK(k2hip_offline_submit_samples(m, smp.data(), n_each, Bp, mt, &t0));
        OK(k2hip_offline_submit_samples(m, smp.data(), n_each, Bp, mt, &t1));
        CHECK(k2hip_offline_submit_samples(m, smp.data(), n_each, Bp, mt, &t2) == K2HIP_ERR_INVALID);
        std::vector<int64_t> tok((size_t)Bp * mt);
        std::vector<int32_t> tsv((size_t)Bp * mt), nn((size_t)Bp);
        OK(k2hip_offline_wait(m, t0, tok.data(), tsv.data(), nn.data()));
        CHECK(k2hip_offline_wait(m, t0, tok.data(), tsv.data(), nn.data()) == K2HIP_ERR_INVALID);   // already collected
        OK(k2hip_offline_wait(m, t1, tok.data(), tsv.data(), nn.data()));
        // operator level
        const int N = 1 + (int)(rng() % 6);
        std::vector<int64_t> y((size_t)N * info.context_size, 1);
        std::vector<float> dec((size_t)N * info.joiner_dim), enc((size_t)N * info.joiner_dim, 0.1f), lg((size_t)N * info.vocab_size);
        OK(k2hip_decoder(m, y.data(), N, dec.data()));
        OK(k2hip_joiner(m, enc.data(), dec.data(), N, lg.data()));
    }
    OK(k2hip_model_destroy(m));
    printf("offline ok: %d rounds, %lld tokens\n", rounds, toks);
    return 0;
}

int errors(const char* path) {
    k2hip_model_t *m = nullptr, *m2 = nullptr;
    CHECK(k2hip_model_create(nullptr, nullptr, 0, &m) == K2HIP_ERR_INVALID);
    CHECK(k2hip_model_create("/nonexistent/model.k2w", nullptr, 0, &m) == K2HIP_ERR_IO && m == nullptr);
    CHECK(k2hip_model_create(path, nullptr, 5, &m) == K2HIP_ERR_NO_DEVICE && m == nullptr);
    OK(k2hip_model_create(path, nullptr, 0, &m));
    OK(k2hip_model_create(path, nullptr, 0, &m2));
    int32_t T = 0, S = 0, Tp = 0;
    OK(k2hip_online_chunk_info(m, &T, &S, &Tp));
    k2hip_online_stream_t *a = nullptr, *b = nullptr, *c = nullptr;
    OK(k2hip_online_stream_create(m, &a));
    OK(k2hip_online_stream_create(m, &b));
    OK(k2hip_online_stream_create(m2, &c));
    std::vector<float> f((size_t)(T + 2 * S) * 80, 2.f);
    OK(k2hip_online_stream_accept_features(a, f.data(), T + 2 * S));
    OK(k2hip_online_stream_accept_features(b, f.data(), T + 2 * S));
    int32_t dec[3], nn[3];
    k2hip_online_stream_t* dup[2] = {a, a};
    CHECK(k2hip_online_step(m, dup, 2, dec, nn) == K2HIP_ERR_INVALID);        // one stream twice
    k2hip_online_stream_t* mixed[2] = {a, c};
    CHECK(k2hip_online_step(m, mixed, 2, dec, nn) == K2HIP_ERR_INVALID);      // a stream of another model
    k2hip_online_stream_t* withnull[2] = {a, nullptr};
    CHECK(k2hip_online_step(m, withnull, 2, dec, nn) == K2HIP_ERR_INVALID);
    CHECK(k2hip_online_step(m, nullptr, 2, dec, nn) == K2HIP_ERR_INVALID);
    k2hip_online_stream_t* both[2] = {a, b};
    // a device failure inside the step: nothing host-side moves, both streams are refused afterwards, reset brings them back
    const long long la = k2hip_online_stream_speech_length(a);
    k2hip_stub_fail_next_step(1);
    CHECK(k2hip_online_step(m, both, 2, dec, nn) == K2HIP_ERR_HIP);
    CHECK(k2hip_online_stream_speech_length(a) == la && k2hip_online_stream_num_tokens(a) == 2);
    CHECK(k2hip_online_step(m, both, 2, dec, nn) == K2HIP_ERR_INVALID);
    CHECK(strstr(k2hip_last_error(), "reset it first") != nullptr);
    OK(k2hip_online_stream_reset(a));
    OK(k2hip_online_stream_reset(b));
    OK(k2hip_online_stream_accept_features(a, f.data(), T));
    k2hip_online_stream_t* one[1] = {a};
    OK(k2hip_online_step(m, one, 1, dec, nn));
    CHECK(dec[0] == 1);
    // the deferred fbank download fails AFTER a step that succeeded (samples were queued: their frames were to be collected behind the
    // step): the stream's FIFO holds placeholder zeros where audio should be -- it must refuse further steps, not decode silence
    {
        OK(k2hip_online_stream_reset(a));
        std::vector<float> pcm((size_t)(T - 1) * 160 + 400 + 160 * (size_t)S, 0.25f);   // a chunk and a shift's worth of samples
        OK(k2hip_online_stream_accept_samples(a, pcm.data(), (int64_t)pcm.size()));
        k2hip_stub_fail_next_gather_finish(1);
        const int32_t rc = k2hip_online_step(m, one, 1, dec, nn);
        fprintf(stderr, "deferred-download failure: step returned %d\n", rc);
        if (rc == K2HIP_ERR_HIP) {   // (the stand-in only defers when the engine would: one batched launch covering every ready stream)
            CHECK(k2hip_online_stream_num_tokens(a) == 2);                         // nothing host-side moved
            CHECK(k2hip_online_step(m, one, 1, dec, nn) == K2HIP_ERR_INVALID);     // poisoned
            CHECK(strstr(k2hip_last_error(), "reset it first") != nullptr);
        } else {
            CHECK(rc == K2HIP_OK);
            k2hip_stub_fail_next_gather_finish(0);
        }
        OK(k2hip_online_stream_reset(a));
        OK(k2hip_online_stream_accept_features(a, f.data(), T));
    }
    // the FIFO mirror: a block larger than the device ring is not mirrored, and is again once it fits
    OK(k2hip_online_stream_reset(b));
    std::vector<float> big((size_t)700 * 80, 3.f);
    OK(k2hip_online_stream_accept_features(b, big.data(), 700));
    int32_t mir = -1;
    OK(k2hip_debug_stream_mirrored(b, &mir));
    CHECK(mir == 0);
    k2hip_online_stream_t* oneb[1] = {b};
    for (int k = 0; k < 6; k++) OK(k2hip_online_step(m, oneb, 1, dec, nn));
    OK(k2hip_debug_stream_mirrored(b, &mir));
    CHECK(mir == 1);
    // operator-level states
    k2hip_online_state_t* st = nullptr;
    OK(k2hip_online_state_create(m, &st));
    std::vector<float> chunk((size_t)T * 80, 1.f), enc((size_t)Tp * 512);
    k2hip_online_state_t* sts[1] = {st};
    CHECK(k2hip_online_encoder(m, sts, 1, chunk.data(), enc.data(), 3) == K2HIP_ERR_CAPACITY);
    OK(k2hip_online_encoder(m, sts, 1, chunk.data(), enc.data(), (int64_t)enc.size()));
    const long long pl = k2hip_online_state_processed_len(st);
    k2hip_stub_fail_next_step(1);
    CHECK(k2hip_online_encoder(m, sts, 1, chunk.data(), enc.data(), (int64_t)enc.size()) == K2HIP_ERR_HIP);
    CHECK(k2hip_online_state_processed_len(st) == pl);
    CHECK(k2hip_online_encoder(m, sts, 1, chunk.data(), enc.data(), (int64_t)enc.size()) == K2HIP_ERR_INVALID);   // poisoned
    OK(k2hip_online_state_destroy(st));
    OK(k2hip_online_stream_destroy(a));
    OK(k2hip_online_stream_destroy(b));
    OK(k2hip_online_stream_destroy(c));
    OK(k2hip_model_destroy(m));
    OK(k2hip_model_destroy(m2));
    printf("errors ok\n");
    return 0;
}

// k2hip_model_create + k2hip_model_meta(key): how csrc/model.cpp derives CustomMetadata values from a container's metadata map
// (prints "<value>" or "ERR <code> <message>")
int meta(const char* path, const char* key) {
    k2hip_model_t* m = nullptr;
    int32_t rc = k2hip_model_create(path, nullptr, 0, &m);
    if (rc) { printf("ERR %d %s\n", rc, k2hip_last_error()); return 0; }
    char buf[512];
    rc = k2hip_model_meta(m, key, buf, sizeof buf);
    if (rc) printf("ERR %d %s\n", rc, k2hip_last_error());
    else printf("%s\n", buf);
    k2hip_model_destroy(m);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "meta")) return meta(argv[2], argv[3]);
    if (argc >= 5 && !strcmp(argv[1], "online")) return online(argv[2], strtoull(argv[3], nullptr, 10), atoi(argv[4]));
    if (argc >= 5 && !strcmp(argv[1], "offline")) return offline(argv[2], strtoull(argv[3], nullptr, 10), atoi(argv[4]));
    if (argc >= 3 && !strcmp(argv[1], "errors")) return errors(argv[2]);
    fprintf(stderr, "usage: san_api_driver online|offline <model.k2w> <seed> <rounds> | errors <streaming.k2w> | meta <model.k2w> <key>\n");
    return 2;
}
