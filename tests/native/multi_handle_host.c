/*
 * multi_handle_host.c -- the shape of the reference-side host on a multi-GPU node, in C: N model handles opened and driven
 * from N host threads (one per GPU; fewer GPUs than threads: the handles share devices), each decoding its own contiguous
 * shard of one utterance list as the GetResults batches shard.py's shard_range / batches_of define, results concatenated in
 * handle order -- and the same list through ONE handle, which must give the same tokens and timestamps, utterance for
 * utterance (a shard is decoded exactly as the reference would decode it as its own batches: OfflineRecognizer.cs:189-303
 * pads and switches context per batch, never across batches).  No collective, no shared state but the result arrays.
 *
 * The N handles are opened and driven the way the C# shim drives them (csharp/OfflineRecognizer.Hip.cs): by MODEL SPEC
 * "model.k2w@device" (k2hip_model_create_spec: the naming K2Hip.SplitSpec parses out of the reference's unchanged constructor
 * arguments), and through native OfflineStreams -- AddSamples = k2hip_offline_stream_accept_samples (raw samples, queued),
 * GetResults = k2hip_offline_recognizer_get_results (fbank + pad + encoder + search on the device), Tokens with the reference's
 * 2 x B blank prefix stripped here.  The ONE-handle decode they are compared with goes through k2hip_offline_greedy_from_samples.
 *
 * usage: multi_handle_host <model.k2w> <samples.f32> <n_utts> <n_samples_each> <batch> <n_handles> <beam (0 = greedy)> <out.bin>
 *   samples.f32: n_utts x n_samples_each little-endian floats.
 *   out.bin: int32 n_utts, int32 max_tokens, then per utterance int32 n, int64 tokens[max_tokens], int32 timestamps[max_tokens]
 * exit code 0 = the two decodes agree.  Built by tests/test_multi_handle_gpu.py: gcc -std=c99 -pthread ... -lk2hip.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "k2hip.h"

typedef struct {
    const char* model_path;
    const float* samples;
    long n_each;
    int total, batch, world, rank, nranks, beam, device, max_tokens;   /* this handle decodes the shards rank .. rank + nranks - 1 */
    int via_streams;   /* 1: spec + OfflineStream route (the C# shim's calls); 0: k2hip_offline_greedy_from_samples */
    int64_t* tokens;   /* [total][max_tokens], shared: every thread writes its own shard's rows */
    int32_t* ts;
    int32_t* n_tok;
    int rc;
    char err[512];
} job_t;

/* shard.py shard_range: contiguous, sizes differ by at most one */
static void shard_range(int n, int world, int rank, int* lo, int* hi) {
    const int base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

static void* run(void* arg) {
    job_t* j = (job_t*)arg;
    k2hip_model_t* m = NULL;
    if (j->via_streams) {
        char spec[1200];
        snprintf(spec, sizeof spec, "%s@%d", j->model_path, j->device);   /* "model.k2w@3": what the C# constructor is handed */
        j->rc = k2hip_model_create_spec(spec, NULL, &m);
    } else {
        j->rc = k2hip_model_create(j->model_path, NULL, j->device, &m);
    }
    if (j->rc) { snprintf(j->err, sizeof j->err, "model_create: %s", k2hip_last_error()); return NULL; }
    {
        k2hip_model_info info;
        if (k2hip_model_get_info(m, &info) || info.device != j->device) {
            snprintf(j->err, sizeof j->err, "the handle is on device %d, not on device %d", info.device, j->device);
            j->rc = -1;
            k2hip_model_destroy(m);
            return NULL;
        }
    }
    if (j->beam > 0 && (j->rc = k2hip_set_decoding_method(m, "modified_beam_search", j->beam))) {
        snprintf(j->err, sizeof j->err, "set_decoding_method: %s", k2hip_last_error());
        k2hip_model_destroy(m);
        return NULL;
    }
    const float** ptrs = (const float**)malloc(sizeof(float*) * (size_t)j->batch);
    int64_t* lens = (int64_t*)malloc(sizeof(int64_t) * (size_t)j->batch);
    for (int r = j->rank; r < j->rank + j->nranks && !j->rc; r++) {
    int lo, hi;
    shard_range(j->total, j->world, r, &lo, &hi);
    for (int a = lo; a < hi && !j->rc; a += j->batch) {   /* shard.py batches_of: a batch never crosses a shard boundary */
        const int cnt = hi - a < j->batch ? hi - a : j->batch;
        for (int i = 0; i < cnt; i++) {
            ptrs[i] = j->samples + (size_t)(a + i) * (size_t)j->n_each;
            lens[i] = j->n_each;
        }
        if (!j->via_streams) {
            j->rc = k2hip_offline_greedy_from_samples(m, ptrs, lens, cnt, j->tokens + (size_t)a * j->max_tokens, j->ts + (size_t)a * j->max_tokens,
                                                      j->n_tok + a, j->max_tokens);
        } else {   /* CreateOfflineStream x cnt, AddSamples in two pieces, GetResults, Tokens / Timestamps behind the 2 x cnt prefix */
            k2hip_offline_stream_t* ss[64];
            for (int i = 0; i < cnt && !j->rc; i++) {
                ss[i] = NULL;
                j->rc = k2hip_offline_stream_create(m, &ss[i]);
                const long half = (j->n_each / 2) | 1;   /* an odd split: the second piece starts inside a frame */
                if (!j->rc) j->rc = k2hip_offline_stream_accept_samples(ss[i], ptrs[i], half);
                if (!j->rc) j->rc = k2hip_offline_stream_accept_samples(ss[i], ptrs[i] + half, j->n_each - half);
            }
            if (!j->rc) j->rc = k2hip_offline_recognizer_get_results(m, ss, cnt);
            for (int i = 0; i < cnt && !j->rc; i++) {
                int64_t tk[4096];
                int32_t tv[4096];
                const int nt = k2hip_offline_stream_num_tokens(ss[i]), nv = k2hip_offline_stream_num_timestamps(ss[i]);
                const int n = nt - 2 * cnt;
                if (n < 0 || n > j->max_tokens || nt > 4096 || nv != nt) { j->rc = -1; snprintf(j->err, sizeof j->err, "stream %d holds %d tokens, %d timestamps", a + i, nt, nv); break; }
                j->rc = k2hip_offline_stream_get_tokens(ss[i], tk, 4096);
                if (!j->rc) j->rc = k2hip_offline_stream_get_timestamps(ss[i], tv, 4096);
                if (j->rc) break;
                for (int k = 0; k < 2 * cnt; k++)
                    if (tk[k] != K2HIP_BLANK_ID || tv[k] != 0) j->rc = -1;   /* OfflineRecognizer.cs:250-267: the prefix */
                memcpy(j->tokens + (size_t)(a + i) * j->max_tokens, tk + 2 * cnt, sizeof(int64_t) * (size_t)n);
                memcpy(j->ts + (size_t)(a + i) * j->max_tokens, tv + 2 * cnt, sizeof(int32_t) * (size_t)n);
                j->n_tok[a + i] = n;
                if (k2hip_offline_stream_speech_length(ss[i]) != 0 && nt > 2) j->rc = -1;   /* RemoveSamples (:294) */
            }
            for (int i = 0; i < cnt; i++)
                if (ss[i]) k2hip_offline_stream_destroy(ss[i]);
        }
        if (j->rc && !j->err[0]) snprintf(j->err, sizeof j->err, "batch at utterance %d: %s", a, k2hip_last_error());
    }
    }
    free(ptrs);
    free(lens);
    k2hip_model_destroy(m);
    return NULL;
}

/* threads = world: one handle and one thread per shard; threads = 1: ONE handle walks the same shards' batches one after the other
 * (the batch list is the same -- the reference's results depend on what shares a GetResults batch, so that is what must not change) */
static int decode(const char* path, const float* samples, long n_each, int total, int batch, int world, int threads, int beam, int ndev,
                  int max_tokens, int64_t* tokens, int32_t* ts, int32_t* n_tok) {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    for (int r = 0; r < threads; r++) {
        job_t* j = &jobs[r];
        j->model_path = path; j->samples = samples; j->n_each = n_each; j->total = total; j->batch = batch; j->world = world;
        j->rank = threads == 1 ? 0 : r; j->nranks = threads == 1 ? world : 1;
        j->beam = beam; j->device = r % ndev; j->max_tokens = max_tokens; j->tokens = tokens; j->ts = ts; j->n_tok = n_tok;
        j->via_streams = threads > 1;
        if (pthread_create(&th[r], NULL, run, j)) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    }
    int rc = 0;
    for (int r = 0; r < threads; r++) {
        pthread_join(th[r], NULL);
        if (jobs[r].rc) { fprintf(stderr, "handle %d (device %d): %s\n", r, jobs[r].device, jobs[r].err); rc = 1; }
    }
    free(th);
    free(jobs);
    return rc;
}

int main(int argc, char** argv) {
    if (argc != 9) {
        fprintf(stderr, "usage: %s <model.k2w> <samples.f32> <n_utts> <n_samples_each> <batch> <n_handles> <beam> <out.bin>\n", argv[0]);
        return 2;
    }
    const char* path = argv[1];
    const int total = atoi(argv[3]), batch = atoi(argv[5]), beam = atoi(argv[7]);
    const long n_each = atol(argv[4]);
    int world = atoi(argv[6]);
    const int ndev = k2hip_device_count();
    if (ndev <= 0) { fprintf(stderr, "no HIP device\n"); return 3; }
    if (world > 8) world = 8;
    if (world < 1 || total < world || batch < 1 || batch > 64 || n_each < 400) { fprintf(stderr, "bad arguments\n"); return 2; }
    const size_t nfl = (size_t)total * (size_t)n_each;
    float* samples = (float*)malloc(sizeof(float) * nfl);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(samples, sizeof(float), nfl, f) != nfl) { fprintf(stderr, "cannot read %zu floats from %s\n", nfl, argv[2]); return 2; }
    fclose(f);
    /* max_tokens: frames of encoder_out for these utterances (a greedy / beam result cannot be longer) */
    k2hip_model_t* probe = NULL;
    if (k2hip_model_create(path, NULL, 0, &probe)) { fprintf(stderr, "model_create: %s\n", k2hip_last_error()); return 1; }
    const int T = (int)k2hip_fbank_num_frames(probe, n_each) + 19;
    const int max_tokens = k2hip_encoder_out_frames(probe, T);
    k2hip_model_destroy(probe);
    if (max_tokens <= 0) { fprintf(stderr, "utterances too short\n"); return 2; }
    const size_t cells = (size_t)total * (size_t)max_tokens;
    int64_t* tokN = (int64_t*)calloc(cells, sizeof(int64_t));
    int64_t* tok1 = (int64_t*)calloc(cells, sizeof(int64_t));
    int32_t* tsN = (int32_t*)calloc(cells, sizeof(int32_t));
    int32_t* ts1 = (int32_t*)calloc(cells, sizeof(int32_t));
    int32_t* nN = (int32_t*)calloc((size_t)total, sizeof(int32_t));
    int32_t* n1 = (int32_t*)calloc((size_t)total, sizeof(int32_t));
    if (decode(path, samples, n_each, total, batch, world, world, beam, ndev, max_tokens, tokN, tsN, nN)) return 1;
    if (decode(path, samples, n_each, total, batch, world, 1, beam, ndev, max_tokens, tok1, ts1, n1)) return 1;
    int bad = 0;
    long emitted = 0;
    for (int u = 0; u < total; u++) {
        emitted += nN[u];
        if (nN[u] != n1[u] || memcmp(tokN + (size_t)u * max_tokens, tok1 + (size_t)u * max_tokens, sizeof(int64_t) * (size_t)nN[u]) ||
            memcmp(tsN + (size_t)u * max_tokens, ts1 + (size_t)u * max_tokens, sizeof(int32_t) * (size_t)nN[u])) {
            if (bad < 5) fprintf(stderr, "utterance %d: %d handles and one handle disagree (%d vs %d tokens)\n", u, world, nN[u], n1[u]);
            bad++;
        }
    }
    f = fopen(argv[8], "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[8]); return 2; }
    const int32_t hdr[2] = {total, max_tokens};
    fwrite(hdr, sizeof(int32_t), 2, f);
    for (int u = 0; u < total; u++) {
        fwrite(&nN[u], sizeof(int32_t), 1, f);
        fwrite(tokN + (size_t)u * max_tokens, sizeof(int64_t), (size_t)max_tokens, f);
        fwrite(tsN + (size_t)u * max_tokens, sizeof(int32_t), (size_t)max_tokens, f);
    }
    fclose(f);
    printf("%d utterances, %d handles on %d device(s), batches of %d, %s: %ld tokens, %d utterances differ from the one-handle decode\n", total, world,
           ndev, batch, beam > 0 ? "modified beam search" : "greedy search", emitted, bad);
    return bad ? 1 : 0;
}
