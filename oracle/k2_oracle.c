/*
 * k2_oracle.c -- CPU restatement of the K2TransducerAsr offline RNN-T hot path.
 * TEST INFRASTRUCTURE ONLY (see k2_oracle.h).  PARITY UNPINNED (see k2_oracle.h).
 *
 * All tensors are row-major f32.  Activations are kept batch-major [B, T, D]
 * (icefall uses [T, B, D]; every op on the path is independent per batch row,
 * so the two are the same function).
 */
#define _GNU_SOURCE
#include "k2_oracle.h"

#include <fcntl.h>
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define MAX_STACKS 8

/* ------------------------------------------------------------------------- */
/* errors                                                                    */
/* ------------------------------------------------------------------------- */
static __thread char g_err[512];
const char* k2o_last_error(void) { return g_err; }
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

/* ------------------------------------------------------------------------- */
/* model container                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    char* name;
    int dtype, ndim;
    int64_t dims[4];
    const void* data;
    float* wt; /* lazily built transpose [K,N] of a 2-D [N,K] f32 tensor */
} tensor_t;

struct k2o_model {
    void* map;
    size_t map_size;
    int n_meta;
    char** keys;
    char** vals;
    int n_tensors;
    tensor_t* t;
    /* parsed */
    int ns;
    int dim[MAX_STACKS], nlayer[MAX_STACKS], ff[MAX_STACKS], heads[MAX_STACKS], kern[MAX_STACKS], ds[MAX_STACKS];
    int qhd[MAX_STACKS], vhd[MAX_STACKS], phd[MAX_STACKS];
    int att[MAX_STACKS]; /* model_type "zipformer" (v1): attention_dims */
    int zip1;            /* model_type "zipformer": streaming Zipformer v1, see k2_oracle_zipformer1.c */
    int pos_dim, J, DD, V, ctx, feat;
    int dmax;
    int conformer; /* model_type "conformer": see k2_oracle_conformer.c */
    int lstm;      /* model_type "lstm": see k2_oracle_lstm.c */
    int rnn_hidden;
    int ctc;       /* model_type "zipformer2ctc": Zipformer2 encoder + CTC head, no decoder / joiner */
    int conv_cpg;  /* decoder conv input channels per group (4: Zipformer recipes; DD: stateless2, groups = 1) */
    /* fbank */
    int sample_rate, frame_len, frame_shift, padded;
    float preemph, low_freq, high_freq, input_scale;
    int remove_dc, snip_edges;
    char window_type[32];
    float* window;   /* [frame_len] */
    float* melw;     /* [feat, padded/2] */
};

const char* k2o_meta(const k2o_model* m, const char* key) {
    for (int i = 0; i < m->n_meta; i++)
        if (!strcmp(m->keys[i], key)) return m->vals[i];
    return NULL;
}
int k2o_vocab_size(const k2o_model* m) { return m->V; }
int k2o_joiner_dim(const k2o_model* m) { return m->J; }
int k2o_encoder_out_dim(const k2o_model* m) { return m->ctc ? m->V : m->J; }
int k2o_context_size(const k2o_model* m) { return m->ctx; }
int k2o_feature_dim(const k2o_model* m) { return m->feat; }

static int parse_csv(const char* s, int* out, int cap) {
    int n = 0;
    if (!s) return 0;
    while (*s && n < cap) {
        out[n++] = (int)strtol(s, (char**)&s, 10);
        if (*s == ',') s++;
    }
    return n;
}
static int meta_int(const k2o_model* m, const char* k, int dflt) {
    const char* v = k2o_meta(m, k);
    return v ? atoi(v) : dflt;
}
static float meta_float(const k2o_model* m, const char* k, float dflt) {
    const char* v = k2o_meta(m, k);
    return v ? (float)atof(v) : dflt;
}

static tensor_t* find_t(const k2o_model* m, const char* name) {
    for (int i = 0; i < m->n_tensors; i++)
        if (!strcmp(m->t[i].name, name)) return &m->t[i];
    return NULL;
}
static const float* W(const k2o_model* m, const char* fmt, ...) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    tensor_t* t = find_t(m, name);
    if (!t) {
        fprintf(stderr, "k2_oracle: missing tensor %s\n", name);
        abort();
    }
    return (const float*)t->data;
}
/* transpose of a Linear weight [N,K] -> [K,N], cached */
static const float* WT(const k2o_model* m, int N, int K, const char* fmt, ...) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    tensor_t* t = find_t(m, name);
    if (!t) {
        fprintf(stderr, "k2_oracle: missing tensor %s\n", name);
        abort();
    }
    int64_t n0 = t->dims[0], k0 = 1;
    for (int i = 1; i < t->ndim; i++) k0 *= t->dims[i];
    if (n0 != N || k0 != K) {
        fprintf(stderr, "k2_oracle: tensor %s is [%ld,%ld], expected [%d,%d]\n", name, (long)n0, (long)k0, N, K);
        abort();
    }
#pragma omp critical(k2o_wt)
    {
        if (!t->wt) {
            float* wt = (float*)malloc(sizeof(float) * (size_t)N * K);
            const float* w = (const float*)t->data;
            for (int n = 0; n < N; n++)
                for (int k = 0; k < K; k++) wt[(size_t)k * N + n] = w[(size_t)n * K + k];
            t->wt = wt;
        }
    }
    return t->wt;
}

static void build_fbank_tables(k2o_model* m);

k2o_model* k2o_model_load(const char* path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        fail("cannot open %s", path);
        return NULL;
    }
    struct stat st;
    fstat(fd, &st);
    void* map = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        fail("mmap failed for %s", path);
        return NULL;
    }
    const uint8_t* p = (const uint8_t*)map;
    if (st.st_size < 24 || memcmp(p, "K2W1", 4)) {
        munmap(map, st.st_size);
        fail("%s: not a K2W1 file", path);
        return NULL;
    }
    uint32_t version, n_meta, n_tensors;
    uint64_t data_off;
    memcpy(&version, p + 4, 4);
    memcpy(&n_meta, p + 8, 4);
    memcpy(&n_tensors, p + 12, 4);
    memcpy(&data_off, p + 16, 8);
    if (version != 1) {
        munmap(map, st.st_size);
        fail("%s: unsupported version %u", path, version);
        return NULL;
    }
    k2o_model* m = (k2o_model*)calloc(1, sizeof *m);
    m->map = map;
    m->map_size = st.st_size;
    m->n_meta = n_meta;
    m->keys = (char**)calloc(n_meta, sizeof(char*));
    m->vals = (char**)calloc(n_meta, sizeof(char*));
    size_t q = 24;
    for (uint32_t i = 0; i < n_meta; i++) {
        uint32_t kl, vl;
        memcpy(&kl, p + q, 4);
        memcpy(&vl, p + q + 4, 4);
        q += 8;
        m->keys[i] = strndup((const char*)p + q, kl);
        q += kl;
        m->vals[i] = strndup((const char*)p + q, vl);
        q += vl;
    }
    m->n_tensors = n_tensors;
    m->t = (tensor_t*)calloc(n_tensors, sizeof(tensor_t));
    for (uint32_t i = 0; i < n_tensors; i++) {
        uint32_t nl, dt, nd;
        uint64_t d[4], off, nb;
        memcpy(&nl, p + q, 4);
        q += 4;
        m->t[i].name = strndup((const char*)p + q, nl);
        q += nl;
        memcpy(&dt, p + q, 4);
        memcpy(&nd, p + q + 4, 4);
        memcpy(d, p + q + 8, 32);
        memcpy(&off, p + q + 40, 8);
        memcpy(&nb, p + q + 48, 8);
        q += 56;
        m->t[i].dtype = dt;
        m->t[i].ndim = nd;
        for (int k = 0; k < 4; k++) m->t[i].dims[k] = (int64_t)d[k];
        m->t[i].data = p + data_off + off;
        (void)nb;
    }
    const char* mt = k2o_meta(m, "model_type");
    m->conformer = mt && !strcmp(mt, "conformer");
    m->ctc = mt && !strcmp(mt, "zipformer2ctc");
    m->lstm = mt && !strcmp(mt, "lstm");
    m->zip1 = mt && !strcmp(mt, "zipformer");
    m->rnn_hidden = meta_int(m, "rnn_hidden_size", 0);
    if (!mt || (strcmp(mt, "zipformer2") && !m->conformer && !m->ctc && !m->lstm && !m->zip1)) {
        fail("model_type %s not supported by the oracle", mt ? mt : "(none)");
        k2o_model_free(m);
        return NULL;
    }
    m->ns = parse_csv(k2o_meta(m, "encoder_dims"), m->dim, MAX_STACKS);
    parse_csv(k2o_meta(m, "num_encoder_layers"), m->nlayer, MAX_STACKS);
    parse_csv(k2o_meta(m, "feedforward_dims"), m->ff, MAX_STACKS);
    parse_csv(k2o_meta(m, "num_heads"), m->heads, MAX_STACKS);
    parse_csv(k2o_meta(m, "cnn_module_kernels"), m->kern, MAX_STACKS);
    parse_csv(k2o_meta(m, "downsampling_factors"), m->ds, MAX_STACKS);
    parse_csv(k2o_meta(m, "query_head_dims"), m->qhd, MAX_STACKS);
    parse_csv(k2o_meta(m, "value_head_dims"), m->vhd, MAX_STACKS);
    parse_csv(k2o_meta(m, "pos_head_dims"), m->phd, MAX_STACKS);
    parse_csv(k2o_meta(m, "attention_dims"), m->att, MAX_STACKS);
    m->pos_dim = meta_int(m, "pos_dim", 48);
    m->J = meta_int(m, "joiner_dim", 512);
    m->DD = meta_int(m, "decoder_dim", 512);
    m->V = meta_int(m, "vocab_size", 500);
    m->ctx = meta_int(m, "context_size", 2);
    m->feat = meta_int(m, "feature_dim", 80);
    {
        tensor_t* cw = find_t(m, "decoder.conv.weight");
        m->conv_cpg = cw ? (int)cw->dims[1] : 4;
    }
    m->dmax = 0;
    for (int i = 0; i < m->ns; i++)
        if (m->dim[i] > m->dmax) m->dmax = m->dim[i];
    m->sample_rate = meta_int(m, "sample_rate", 16000);
    m->frame_len = m->sample_rate * meta_int(m, "frame_length_ms", 25) / 1000;
    m->frame_shift = m->sample_rate * meta_int(m, "frame_shift_ms", 10) / 1000;
    m->padded = 1;
    while (m->padded < m->frame_len) m->padded <<= 1;
    m->preemph = meta_float(m, "preemph_coeff", 0.97f);
    m->low_freq = meta_float(m, "low_freq", 20.f);
    m->high_freq = meta_float(m, "high_freq", 0.f);
    m->input_scale = meta_float(m, "input_scale", 1.f);
    m->remove_dc = meta_int(m, "remove_dc_offset", 1);
    m->snip_edges = meta_int(m, "snip_edges", 1);
    const char* wt = k2o_meta(m, "window_type");
    snprintf(m->window_type, sizeof m->window_type, "%s", wt ? wt : "hamming");
    build_fbank_tables(m);
    return m;
}

void k2o_model_free(k2o_model* m) {
    if (!m) return;
    for (int i = 0; i < m->n_meta; i++) {
        free(m->keys[i]);
        free(m->vals[i]);
    }
    free(m->keys);
    free(m->vals);
    for (int i = 0; i < m->n_tensors; i++) {
        free(m->t[i].name);
        free(m->t[i].wt);
    }
    free(m->t);
    free(m->window);
    free(m->melw);
    if (m->map) munmap(m->map, m->map_size);
    free(m);
}

/* ------------------------------------------------------------------------- */
/* F1: kaldi-style fbank.  The reference calls SpeechFeatures.OnlineFbank     */
/* (WavFrontend.cs:22-35) with dither=0, snip_edges=true, 16 kHz, 80 bins,    */
/* window "hamming" (Model/FrontendConfEntity.cs:8), feature_type "fbank";    */
/* that package's source is not in the reference, so the body follows the     */
/* published kaldi / kaldi-native-fbank algorithm (feature-window.cc,         */
/* mel-computations.cc, feature-fbank.cc) with its defaults.                  */
/* ------------------------------------------------------------------------- */
/* kaldi MelBanks uses MelScale(f) = 1127 ln(1 + f/700) and triangular filters in the
 * mel domain.  Evaluated in f64 and rounded once to f32: in f32 the subtraction
 * mel - left cancels ~4 digits, so the weights would depend on the compiler's
 * contraction choices at the 1e-6 level. */
static double mel_scale(double f) { return 1127.0 * log(1.0 + f / 700.0); }

static void build_fbank_tables(k2o_model* m) {
    int N = m->frame_len;
    m->window = (float*)malloc(sizeof(float) * N);
    double a = 2.0 * M_PI / (N - 1);
    for (int i = 0; i < N; i++) {
        double w;
        if (!strcmp(m->window_type, "hamming")) w = 0.54 - 0.46 * cos(a * i);
        else if (!strcmp(m->window_type, "hanning")) w = 0.5 - 0.5 * cos(a * i);
        else if (!strcmp(m->window_type, "povey")) w = pow(0.5 - 0.5 * cos(a * i), 0.85);
        else if (!strcmp(m->window_type, "rectangular")) w = 1.0;
        else w = 0.54 - 0.46 * cos(a * i);
        m->window[i] = (float)w;
    }
    int nb = m->padded / 2;
    m->melw = (float*)calloc((size_t)m->feat * nb, sizeof(float));
    double nyq = 0.5 * m->sample_rate;
    double hi = m->high_freq;
    if (hi <= 0.0) hi += nyq;
    double fft_bin_width = (double)m->sample_rate / m->padded;
    double mel_low = mel_scale(m->low_freq), mel_high = mel_scale(hi);
    double delta = (mel_high - mel_low) / (m->feat + 1);
    for (int b = 0; b < m->feat; b++) {
        double left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
        for (int i = 0; i < nb; i++) {
            double mel = mel_scale(fft_bin_width * i);
            if (mel > left && mel < right) {
                double w = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
                m->melw[(size_t)b * nb + i] = (float)w;
            }
        }
    }
}

int64_t k2o_fbank_num_frames(const k2o_model* m, int64_t n) {
    if (n < m->frame_len) return 0;
    return 1 + (n - m->frame_len) / m->frame_shift;
}

/* in-place iterative radix-2 complex FFT, n power of two.  f64: the frame is
 * windowed in f32 as kaldi does, but transform, power spectrum and mel sums are
 * carried in f64 so that the oracle is the round-off-free statement of the
 * algorithm (a f32 FFT carries ~1e-4 relative noise in low-energy bins). */
static void fft_c2c(double* re, double* im, int n) {
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; k++) {
                double wr = cos(ang * k), wi = sin(ang * k);
                int a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

int64_t k2o_fbank(const k2o_model* m, const float* samples, int64_t n, float* feats, int64_t cap) {
    int64_t nf = k2o_fbank_num_frames(m, n);
    if (nf > cap) return fail("fbank: %ld frames exceed capacity %ld", (long)nf, (long)cap);
    int N = m->frame_len, P = m->padded, nb = P / 2;
#pragma omp parallel
    {
        double* re = (double*)malloc(sizeof(double) * P);
        double* im = (double*)malloc(sizeof(double) * P);
        double* pw = (double*)malloc(sizeof(double) * (nb + 1));
#pragma omp for schedule(static)
        for (int64_t f = 0; f < nf; f++) {
            const float* s = samples + f * m->frame_shift;
            /* kaldi ProcessWindow order (remove DC, pre-emphasis, window), carried in f64 */
            for (int i = 0; i < N; i++) re[i] = (double)s[i] * (double)m->input_scale;
            if (m->remove_dc) {
                double sum = 0.0;
                for (int i = 0; i < N; i++) sum += re[i];
                double mean = sum / N;
                for (int i = 0; i < N; i++) re[i] -= mean;
            }
            if (m->preemph != 0.f) {
                for (int i = N - 1; i > 0; i--) re[i] -= (double)m->preemph * re[i - 1];
                re[0] -= (double)m->preemph * re[0];
            }
            for (int i = 0; i < N; i++) re[i] *= (double)m->window[i];
            for (int i = N; i < P; i++) re[i] = 0.0;
            for (int i = 0; i < P; i++) im[i] = 0.0;
            fft_c2c(re, im, P);
            for (int i = 0; i <= nb; i++) pw[i] = re[i] * re[i] + im[i] * im[i];
            for (int b = 0; b < m->feat; b++) {
                const float* w = m->melw + (size_t)b * nb;
                double e = 0.0;
                for (int i = 0; i < nb; i++) e += (double)w[i] * pw[i];
                float ef = (float)e;
                if (ef < FLT_EPSILON) ef = FLT_EPSILON;
                feats[f * m->feat + b] = logf(ef);
            }
        }
        free(re); free(im); free(pw);
    }
    return nf;
}

/* ------------------------------------------------------------------------- */
/* F3: PadHelper.PadSequence(List<OfflineInputEntity>) -- PadHelper.cs:14-60  */
/*  Q1: tail is 80*19 floats whatever featureDim is (:17,:22)                 */
/*  Q2: after padding EVERY element == 0 becomes -23.025850929940457f (:58)   */
/* ------------------------------------------------------------------------- */
int64_t k2o_pad_sequence(const float* const* speech, const int64_t* n_floats, int B, int tail_frames, float* out) {
    int64_t mx = 0;
    for (int i = 0; i < B; i++)
        if (n_floats[i] > mx) mx = n_floats[i];
    int64_t L = mx + 80 * (int64_t)tail_frames;
    if (!out) return L;
    for (int i = 0; i < B; i++) {
        float* row = out + (int64_t)i * L;
        for (int64_t j = 0; j < L; j++) {
            float v = (j < n_floats[i]) ? speech[i][j] : 0.0f;
            row[j] = (v == 0.0f) ? -23.025850929940457F : v;
        }
    }
    return L;
}

/* ------------------------------------------------------------------------- */
/* primitive ops                                                             */
/* ------------------------------------------------------------------------- */
static float* falloc(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, sizeof(float) * (n ? n : 1))) abort();
    return (float*)p;
}

/* y[M,N] (ld ldy) = x[M,K] (ld ldx) . wt[K,N] + b ; k summed in ascending order */
static void linear(float* y, int ldy, const float* x, int ldx, const float* wt, const float* b, int M, int K, int N) {
#pragma omp parallel for schedule(static)
    for (int m0 = 0; m0 < M; m0 += 8) {
        int mr = M - m0 < 8 ? M - m0 : 8;
        for (int n0 = 0; n0 < N; n0 += 512) {
            int nn = N - n0 < 512 ? N - n0 : 512;
            float acc[8][512];
            for (int r = 0; r < mr; r++)
                for (int n = 0; n < nn; n++) acc[r][n] = b ? b[n0 + n] : 0.f;
            for (int k = 0; k < K; k++) {
                const float* wr = wt + (size_t)k * N + n0;
                for (int r = 0; r < mr; r++) {
                    float a = x[(size_t)(m0 + r) * ldx + k];
                    float* ar = acc[r];
                    for (int n = 0; n < nn; n++) ar[n] += a * wr[n];
                }
            }
            for (int r = 0; r < mr; r++) memcpy(y + (size_t)(m0 + r) * ldy + n0, acc[r], sizeof(float) * nn);
        }
    }
}

/* scaling.py SwooshL / SwooshR (icefall):  logaddexp(0, x - o) - 0.08 x - c */
static inline float logaddexp0(float z) { return (z > 0.f ? z : 0.f) + log1pf(expf(-fabsf(z))); }
static inline float swoosh_l(float x) { return logaddexp0(x - 4.0f) - 0.08f * x - 0.035f; }
static inline float swoosh_r(float x) { return logaddexp0(x - 1.0f) - 0.08f * x - 0.313261687f; }

static void apply_swoosh_l(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) x[i] = swoosh_l(x[i]);
}
static void apply_swoosh_r(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) x[i] = swoosh_r(x[i]);
}

/* BiasNorm (icefall zipformer.py):  x * (mean((x-bias)^2))^-0.5 * exp(log_scale) */
static void biasnorm(float* y, const float* x, const float* bias, float log_scale, int M, int D) {
    float es = expf(log_scale);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++) {
        const float* xr = x + (size_t)m * D;
        float s = 0.f;
        for (int d = 0; d < D; d++) {
            float v = xr[d] - bias[d];
            s += v * v;
        }
        float sc = (1.0f / sqrtf(s / D)) * es;
        float* yr = y + (size_t)m * D;
        for (int d = 0; d < D; d++) yr[d] = xr[d] * sc;
    }
}

/* BypassModule: orig + (x - orig) * scale[d] */
static void bypass(float* out, const float* orig, const float* x, const float* scale, int M, int D) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++)
        for (int d = 0; d < D; d++) {
            size_t i = (size_t)m * D + d;
            out[i] = orig[i] + (x[i] - orig[i]) * scale[d];
        }
}

static void add_inplace(float* a, const float* b, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) a[i] += b[i];
}

/* ------------------------------------------------------------------------- */
/* encoder_embed: icefall Conv2dSubsampling (subsampling.py), inference graph */
/*  conv(1->8,k3,pad(0,1)) SwooshR; conv(8->32,k3,s2) SwooshR;               */
/*  conv(32->128,k3,s(1,2)) SwooshR; ConvNeXt(128, 7x7); Linear; BiasNorm     */
/*  Activations kept NHWC: [B, T, F, C].                                      */
/* ------------------------------------------------------------------------- */
static int embed_out_frames(int T) { return (T - 7) / 2; }

static float* encoder_embed(const k2o_model* m, const float* x, int B, int T, int* T_out) {
    int F0 = m->feat;
    int T1 = T - 2, F1 = F0;               /* conv0: pad (0,1) */
    int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
    int T3 = T2 - 2, F3 = (F2 - 3) / 2 + 1;
    if (T3 <= 0) { *T_out = 0; return NULL; }
    const float* w0 = W(m, "encoder_embed.conv.0.weight");
    const float* b0 = W(m, "encoder_embed.conv.0.bias");
    float* a1 = falloc((size_t)B * T1 * F1 * 8);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T1; t++)
            for (int f = 0; f < F1; f++)
                for (int co = 0; co < 8; co++) {
                    float s = b0[co];
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++) {
                            int ff = f + kf - 1;
                            if (ff < 0 || ff >= F0) continue;
                            s += w0[(co * 3 + kt) * 3 + kf] * x[((size_t)b * T + t + kt) * F0 + ff];
                        }
                    a1[(((size_t)b * T1 + t) * F1 + f) * 8 + co] = swoosh_r(s);
                }
    const float* w4 = W(m, "encoder_embed.conv.4.weight"); /* [32,8,3,3] */
    const float* b4 = W(m, "encoder_embed.conv.4.bias");
    float* a2 = falloc((size_t)B * T2 * F2 * 32);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T2; t++)
            for (int f = 0; f < F2; f++)
                for (int co = 0; co < 32; co++) {
                    float s = b4[co];
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++) {
                            const float* xi = a1 + (((size_t)b * T1 + 2 * t + kt) * F1 + 2 * f + kf) * 8;
                            for (int ci = 0; ci < 8; ci++) s += w4[((co * 8 + ci) * 3 + kt) * 3 + kf] * xi[ci];
                        }
                    a2[(((size_t)b * T2 + t) * F2 + f) * 32 + co] = swoosh_r(s);
                }
    free(a1);
    const float* w7 = W(m, "encoder_embed.conv.7.weight"); /* [128,32,3,3] */
    const float* b7 = W(m, "encoder_embed.conv.7.bias");
    float* a3 = falloc((size_t)B * T3 * F3 * 128);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T3; t++)
            for (int f = 0; f < F3; f++)
                for (int co = 0; co < 128; co++) {
                    float s = b7[co];
                    for (int kt = 0; kt < 3; kt++)
                        for (int kf = 0; kf < 3; kf++) {
                            const float* xi = a2 + (((size_t)b * T2 + t + kt) * F2 + 2 * f + kf) * 32;
                            for (int ci = 0; ci < 32; ci++) s += w7[((co * 32 + ci) * 3 + kt) * 3 + kf] * xi[ci];
                        }
                    a3[(((size_t)b * T3 + t) * F3 + f) * 128 + co] = swoosh_r(s);
                }
    free(a2);
    /* ConvNeXt: bypass + pw2(SwooshL(pw1(dw7x7(x)))) */
    const float* wd = W(m, "encoder_embed.convnext.depthwise_conv.weight"); /* [128,1,7,7] */
    const float* bd = W(m, "encoder_embed.convnext.depthwise_conv.bias");
    size_t npix = (size_t)B * T3 * F3;
    float* dw = falloc(npix * 128);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T3; t++)
            for (int f = 0; f < F3; f++)
                for (int c = 0; c < 128; c++) {
                    float s = bd[c];
                    for (int kt = 0; kt < 7; kt++) {
                        int tt = t + kt - 3;
                        if (tt < 0 || tt >= T3) continue;
                        for (int kf = 0; kf < 7; kf++) {
                            int ff = f + kf - 3;
                            if (ff < 0 || ff >= F3) continue;
                            s += wd[(c * 7 + kt) * 7 + kf] * a3[(((size_t)b * T3 + tt) * F3 + ff) * 128 + c];
                        }
                    }
                    dw[(((size_t)b * T3 + t) * F3 + f) * 128 + c] = s;
                }
    float* h = falloc(npix * 384);
    linear(h, 384, dw, 128, WT(m, 384, 128, "encoder_embed.convnext.pointwise_conv1.weight"),
           W(m, "encoder_embed.convnext.pointwise_conv1.bias"), (int)npix, 128, 384);
    apply_swoosh_l(h, npix * 384);
    linear(dw, 128, h, 384, WT(m, 128, 384, "encoder_embed.convnext.pointwise_conv2.weight"),
           W(m, "encoder_embed.convnext.pointwise_conv2.bias"), (int)npix, 384, 128);
    free(h);
    add_inplace(a3, dw, npix * 128);
    free(dw);
    /* x.transpose(1,2).reshape(b,t,c*f): feature index = c*F3 + f */
    int D0 = m->dim[0], KK = 128 * F3;
    float* flat = falloc((size_t)B * T3 * KK);
#pragma omp parallel for schedule(static)
    for (int bt = 0; bt < B * T3; bt++)
        for (int f = 0; f < F3; f++)
            for (int c = 0; c < 128; c++) flat[(size_t)bt * KK + c * F3 + f] = a3[((size_t)bt * F3 + f) * 128 + c];
    free(a3);
    float* lin = falloc((size_t)B * T3 * D0);
    linear(lin, D0, flat, KK, WT(m, D0, KK, "encoder_embed.out.weight"), W(m, "encoder_embed.out.bias"), B * T3, KK, D0);
    free(flat);
    float* out = falloc((size_t)B * T3 * D0);
    biasnorm(out, lin, W(m, "encoder_embed.out_norm.bias"), W(m, "encoder_embed.out_norm.log_scale")[0], B * T3, D0);
    free(lin);
    *T_out = T3;
    return out;
}

/* ------------------------------------------------------------------------- */
/* Zipformer2 pieces (icefall zipformer.py, inference / tracing branches)     */
/* ------------------------------------------------------------------------- */

/* CompactRelPositionalEncoding.extend_pe: pe [2T-1, pos_dim], row n <-> offset n-(T-1) */
static float* compact_rel_pos(int T, int pos_dim) {
    int n2 = 2 * T - 1;
    float* pe = falloc((size_t)n2 * pos_dim);
    float cl = sqrtf((float)pos_dim);
    float length_scale = (float)pos_dim / (2.0f * (float)M_PI);
    float logcl = logf(cl);
    for (int n = 0; n < n2; n++) {
        float x = (float)(n - (T - 1));
        float sgn = (x > 0.f) - (x < 0.f);
        float xc = cl * sgn * (logf(fabsf(x) + cl) - logcl);
        float xa = atanf(xc / length_scale);
        for (int k = 0; k < pos_dim / 2; k++) {
            float fr = (float)(k + 1);
            pe[(size_t)n * pos_dim + 2 * k] = cosf(xa * fr);
            pe[(size_t)n * pos_dim + 2 * k + 1] = sinf(xa * fr);
        }
        pe[(size_t)n * pos_dim + pos_dim - 1] = 1.0f;
    }
    return pe;
}

/* RelPositionMultiheadAttentionWeights.forward -> attn [H, B, T, T] */
static float* attn_weights(const k2o_model* m, int si, const char* pfx, const float* src, const float* pe, int B, int T) {
    int D = m->dim[si], H = m->heads[si], q = m->qhd[si], p = m->phd[si];
    int inproj = (2 * q + p) * H, M = B * T, n2 = 2 * T - 1;
    float* x = falloc((size_t)M * inproj);
    linear(x, inproj, src, D, WT(m, inproj, D, "%sself_attn_weights.in_proj.weight", pfx),
           W(m, "%sself_attn_weights.in_proj.bias", pfx), M, D, inproj);
    float* pp = falloc((size_t)n2 * p * H); /* linear_pos(pos_emb): [n2, H*p] */
    linear(pp, p * H, pe, m->pos_dim, WT(m, p * H, m->pos_dim, "%sself_attn_weights.linear_pos.weight", pfx), NULL, n2,
           m->pos_dim, p * H);
    float* aw = falloc((size_t)H * B * T * T);
#pragma omp parallel for collapse(2) schedule(static)
    for (int h = 0; h < H; h++)
        for (int b = 0; b < B; b++)
            for (int i = 0; i < T; i++) {
                const float* qi = x + ((size_t)b * T + i) * inproj + h * q;
                const float* pi = x + ((size_t)b * T + i) * inproj + 2 * q * H + h * p;
                float* row = aw + (((size_t)h * B + b) * T + i) * T;
                float mx = -INFINITY;
                for (int j = 0; j < T; j++) {
                    const float* kj = x + ((size_t)b * T + j) * inproj + q * H + h * q;
                    float s = 0.f;
                    for (int d = 0; d < q; d++) s += qi[d] * kj[d];
                    /* pos_scores[i, j] = (p_i . pos[T-1-i+j])  (gather form of the rel-shift) */
                    const float* pr = pp + (size_t)(T - 1 - i + j) * (p * H) + h * p;
                    float ps = 0.f;
                    for (int c = 0; c < p; c++) ps += pi[c] * pr[c];
                    s += ps;
                    row[j] = s;
                    if (s > mx) mx = s;
                }
                float sum = 0.f;
                for (int j = 0; j < T; j++) {
                    row[j] = expf(row[j] - mx);
                    sum += row[j];
                }
                float inv = 1.0f / sum;
                for (int j = 0; j < T; j++) row[j] *= inv;
            }
    free(x);
    free(pp);
    return aw;
}

/* out[b,i,c0+c] = sum_j aw[b,i,j] * v[b,j,c0+c]   (aw: [B,T,T] of one head) */
static void attn_apply(float* out, int ldo, const float* aw, const float* v, int ldv, int B, int T, int c0, int nc) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int i = 0; i < T; i++) {
            float acc[1024];
            for (int c = 0; c < nc; c++) acc[c] = 0.f;
            const float* a = aw + ((size_t)b * T + i) * T;
            for (int j = 0; j < T; j++) {
                float w = a[j];
                const float* vr = v + ((size_t)b * T + j) * ldv + c0;
                for (int c = 0; c < nc; c++) acc[c] += w * vr[c];
            }
            memcpy(out + ((size_t)b * T + i) * ldo + c0, acc, sizeof(float) * nc);
        }
}

/* src += FeedforwardModule(src) : out_proj(SwooshL(in_proj(x))) */
static void feed_forward(const k2o_model* m, const char* pfx, int k, float* src, int M, int D, int F) {
    float* h = falloc((size_t)M * F);
    linear(h, F, src, D, WT(m, F, D, "%sfeed_forward%d.in_proj.weight", pfx, k), W(m, "%sfeed_forward%d.in_proj.bias", pfx, k), M, D, F);
    apply_swoosh_l(h, (size_t)M * F);
    float* o = falloc((size_t)M * D);
    linear(o, D, h, F, WT(m, D, F, "%sfeed_forward%d.out_proj.weight", pfx, k), W(m, "%sfeed_forward%d.out_proj.bias", pfx, k), M, F, D);
    add_inplace(src, o, (size_t)M * D);
    free(h);
    free(o);
}

/* src += NonlinAttention(src, attn_weights[0:1]) */
static void nonlin_attention(const k2o_model* m, const char* pfx, float* src, const float* aw0, int B, int T, int D) {
    int Hc = 3 * D / 4, M = B * T;
    float* x = falloc((size_t)M * 3 * Hc);
    linear(x, 3 * Hc, src, D, WT(m, 3 * Hc, D, "%snonlin_attention.in_proj.weight", pfx), W(m, "%snonlin_attention.in_proj.bias", pfx), M, D, 3 * Hc);
    float* g = falloc((size_t)M * Hc);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++)
        for (int c = 0; c < Hc; c++) g[(size_t)r * Hc + c] = x[(size_t)r * 3 * Hc + Hc + c] * tanhf(x[(size_t)r * 3 * Hc + c]);
    float* a = falloc((size_t)M * Hc);
    if (Hc > 1024) abort();
    attn_apply(a, Hc, aw0, g, Hc, B, T, 0, Hc);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++)
        for (int c = 0; c < Hc; c++) a[(size_t)r * Hc + c] *= x[(size_t)r * 3 * Hc + 2 * Hc + c];
    float* o = falloc((size_t)M * D);
    linear(o, D, a, Hc, WT(m, D, Hc, "%snonlin_attention.out_proj.weight", pfx), W(m, "%snonlin_attention.out_proj.bias", pfx), M, Hc, D);
    add_inplace(src, o, (size_t)M * D);
    free(x); free(g); free(a); free(o);
}

/* src += SelfAttention_k(src, attn_weights) */
static void self_attn(const k2o_model* m, int si, const char* pfx, int k, float* src, const float* aw, int B, int T) {
    int D = m->dim[si], H = m->heads[si], v = m->vhd[si], M = B * T, HV = H * v;
    float* x = falloc((size_t)M * HV);
    linear(x, HV, src, D, WT(m, HV, D, "%sself_attn%d.in_proj.weight", pfx, k), W(m, "%sself_attn%d.in_proj.bias", pfx, k), M, D, HV);
    float* a = falloc((size_t)M * HV);
    for (int h = 0; h < H; h++) attn_apply(a, HV, aw + (size_t)h * B * T * T, x, HV, B, T, h * v, v);
    float* o = falloc((size_t)M * D);
    linear(o, D, a, HV, WT(m, D, HV, "%sself_attn%d.out_proj.weight", pfx, k), W(m, "%sself_attn%d.out_proj.bias", pfx, k), M, HV, D);
    add_inplace(src, o, (size_t)M * D);
    free(x); free(a); free(o);
}

/* src += ConvolutionModule_k(src): in_proj -> GLU -> depthwise conv1d(K, pad K/2) -> SwooshR -> out_proj */
static void conv_module(const k2o_model* m, int si, const char* pfx, int k, float* src, int B, int T) {
    int D = m->dim[si], K = m->kern[si], M = B * T;
    float* x = falloc((size_t)M * 2 * D);
    linear(x, 2 * D, src, D, WT(m, 2 * D, D, "%sconv_module%d.in_proj.weight", pfx, k), W(m, "%sconv_module%d.in_proj.bias", pfx, k), M, D, 2 * D);
    float* g = falloc((size_t)M * D);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++)
        for (int c = 0; c < D; c++) {
            float s = x[(size_t)r * 2 * D + D + c];
            g[(size_t)r * D + c] = x[(size_t)r * 2 * D + c] * (1.0f / (1.0f + expf(-s)));
        }
    const float* dw = W(m, "%sconv_module%d.depthwise_conv.weight", pfx, k); /* [D,1,K] */
    const float* db = W(m, "%sconv_module%d.depthwise_conv.bias", pfx, k);
    float* c1 = falloc((size_t)M * D);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++)
            for (int c = 0; c < D; c++) {
                float s = db[c];
                for (int kk = 0; kk < K; kk++) {
                    int tt = t + kk - K / 2;
                    if (tt < 0 || tt >= T) continue;
                    s += dw[c * K + kk] * g[((size_t)b * T + tt) * D + c];
                }
                c1[((size_t)b * T + t) * D + c] = swoosh_r(s);
            }
    float* o = falloc((size_t)M * D);
    linear(o, D, c1, D, WT(m, D, D, "%sconv_module%d.out_proj.weight", pfx, k), W(m, "%sconv_module%d.out_proj.bias", pfx, k), M, D, D);
    add_inplace(src, o, (size_t)M * D);
    free(x); free(g); free(c1); free(o);
}

/* Zipformer2EncoderLayer.forward (inference) -- in place on src [B,T,D] */
static void encoder_layer(const k2o_model* m, int si, int li, float* src, const float* pe, int B, int T) {
    char pfx[128];
    snprintf(pfx, sizeof pfx, "encoder.encoders.%d.layers.%d.", si, li);
    int D = m->dim[si], F = m->ff[si], M = B * T;
    size_t n = (size_t)M * D;
    float* orig = falloc(n);
    memcpy(orig, src, sizeof(float) * n);
    float* aw = attn_weights(m, si, pfx, src, pe, B, T);
    feed_forward(m, pfx, 1, src, M, D, F * 3 / 4);
    nonlin_attention(m, pfx, src, aw, B, T, D);
    self_attn(m, si, pfx, 1, src, aw, B, T);
    conv_module(m, si, pfx, 1, src, B, T);
    feed_forward(m, pfx, 2, src, M, D, F);
    bypass(src, orig, src, W(m, "%sbypass_mid.bypass_scale", pfx), M, D);
    self_attn(m, si, pfx, 2, src, aw, B, T);
    conv_module(m, si, pfx, 2, src, B, T);
    feed_forward(m, pfx, 3, src, M, D, F * 5 / 4);
    float* nm = falloc(n);
    biasnorm(nm, src, W(m, "%snorm.bias", pfx), W(m, "%snorm.log_scale", pfx)[0], M, D);
    bypass(src, orig, nm, W(m, "%sbypass.bypass_scale", pfx), M, D);
    free(nm); free(aw); free(orig);
}

/* SimpleDownsample: softmax(bias)-weighted sum of ds consecutive frames, last frame repeated as pad */
static float* simple_downsample(const float* src, const float* bias, int ds, int B, int T, int D, int* Tout) {
    int Td = (T + ds - 1) / ds;
    float w[16], mx = -INFINITY, sum = 0.f;
    for (int i = 0; i < ds; i++) if (bias[i] > mx) mx = bias[i];
    for (int i = 0; i < ds; i++) { w[i] = expf(bias[i] - mx); sum += w[i]; }
    for (int i = 0; i < ds; i++) w[i] /= sum;
    float* out = falloc((size_t)B * Td * D);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < Td; t++)
            for (int d = 0; d < D; d++) {
                float s = 0.f;
                for (int i = 0; i < ds; i++) {
                    int tt = t * ds + i;
                    if (tt >= T) tt = T - 1;
                    s += src[((size_t)b * T + tt) * D + d] * w[i];
                }
                out[((size_t)b * Td + t) * D + d] = s;
            }
    *Tout = Td;
    return out;
}

/* convert_num_channels: truncate or zero-pad the channel dim */
static float* convert_channels(const float* x, int M, int Din, int Dout) {
    float* y = falloc((size_t)M * Dout);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < M; r++)
        for (int d = 0; d < Dout; d++) y[(size_t)r * Dout + d] = d < Din ? x[(size_t)r * Din + d] : 0.f;
    return y;
}

#include "k2_oracle_conformer.c"
#include "k2_oracle_lstm.c"

static int zip1_offline_forward(const k2o_model* m, const float* xin, int B, int T, float* enc_out, int tap, float* tap_out, int64_t tap_cap,
                                int64_t* tap_n);   /* k2_oracle_zipformer1.c */

int k2o_encoder_out_frames(const k2o_model* m, int T) {
    if (m->conformer) return conformer_out_frames(T);
    if (m->lstm) return lstm_out_frames(T);
    int T50 = embed_out_frames(T);
    if (T50 <= 0) return 0;
    return (T50 + 1) / 2;
}

/* CTC head of a zipformer2ctc export (icefall: ctc_output = Dropout, Linear(max(encoder_dims), vocab), LogSoftmax) on the
 * 25 Hz encoder output; this is what OfflineProjOfZipformer2ctc.EncoderProj returns as "log_probs" [B,T',V]
 * (OfflineProjOfZipformer2ctc.cs:48-92, consumed at OfflineRecognizer.cs:323-326). */
static void ctc_head(const k2o_model* m, const float* x, int rows, float* out) {
    const int V = m->V, D = m->dmax;
    linear(out, V, x, D, WT(m, V, D, "ctc_output.1.weight"), W(m, "ctc_output.1.bias"), rows, D, V);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        float* l = out + (size_t)r * V;
        float mx = l[0];
        for (int v = 1; v < V; v++) mx = l[v] > mx ? l[v] : mx;
        float s = 0.f;
        for (int v = 0; v < V; v++) s += expf(l[v] - mx);
        float lse = logf(s);
        for (int v = 0; v < V; v++) l[v] = l[v] - mx - lse;
    }
}

/* Zipformer2.forward + encoder_proj.  taps: see header. */
static int encoder_forward(const k2o_model* m, const float* xin, int B, int T, float* enc_out, int tap, float* tap_out,
                           int64_t tap_cap, int64_t* tap_n) {
    int T50;
    float* x = encoder_embed(m, xin, B, T, &T50);
    if (!x) return fail("encoder: T=%d too short", T);
    int M = B * T50;
    if (tap == 0) {
        int64_t n = (int64_t)M * m->dim[0];
        if (n > tap_cap) { free(x); return fail("tap buffer too small"); }
        memcpy(tap_out, x, sizeof(float) * n);
        *tap_n = n;
        free(x);
        return 0;
    }
    float* outputs[MAX_STACKS] = {0};
    int Dcur = m->dim[0];
    for (int si = 0; si < m->ns; si++) {
        int D = m->dim[si], ds = m->ds[si];
        float* xi = convert_channels(x, M, Dcur, D);
        free(x);
        Dcur = D;
        if (ds == 1) {
            float* pe = compact_rel_pos(T50, m->pos_dim);
            for (int li = 0; li < m->nlayer[si]; li++) encoder_layer(m, si, li, xi, pe, B, T50);
            free(pe);
            x = xi;
        } else {
            /* DownsampledZipformer2Encoder */
            int Td;
            float* xd = simple_downsample(xi, W(m, "encoder.encoders.%d.downsample.bias", si), ds, B, T50, D, &Td);
            float* pe = compact_rel_pos(Td, m->pos_dim);
            for (int li = 0; li < m->nlayer[si]; li++) encoder_layer(m, si, li, xd, pe, B, Td);
            free(pe);
            /* SimpleUpsample (repeat) + truncate + out_combiner(src_orig, src) */
            const float* sc = W(m, "encoder.encoders.%d.out_combiner.bypass_scale", si);
            float* y = falloc((size_t)M * D);
#pragma omp parallel for collapse(2) schedule(static)
            for (int b = 0; b < B; b++)
                for (int t = 0; t < T50; t++)
                    for (int d = 0; d < D; d++) {
                        float o = xi[((size_t)b * T50 + t) * D + d];
                        float u = xd[((size_t)b * Td + t / ds) * D + d];
                        y[((size_t)b * T50 + t) * D + d] = o + (u - o) * sc[d];
                    }
            free(xd);
            free(xi);
            x = y;
        }
        outputs[si] = falloc((size_t)M * D);
        memcpy(outputs[si], x, sizeof(float) * (size_t)M * D);
        if (tap == 1 + si) {
            int64_t n = (int64_t)M * D;
            int rc = 0;
            if (n > tap_cap) rc = fail("tap buffer too small");
            else { memcpy(tap_out, x, sizeof(float) * n); *tap_n = n; }
            for (int k = 0; k <= si; k++) free(outputs[k]);
            free(x);
            return rc;
        }
    }
    free(x);
    /* _get_full_dim_output: last output, then missing channel ranges from earlier, wider outputs */
    int Dmax = m->dmax;
    float* full = falloc((size_t)M * Dmax);
    {
        int cur = m->dim[m->ns - 1];
        for (int r = 0; r < M; r++) memcpy(full + (size_t)r * Dmax, outputs[m->ns - 1] + (size_t)r * cur, sizeof(float) * cur);
        for (int i = m->ns - 2; i >= 0; i--) {
            int d = m->dim[i];
            if (d > cur) {
                for (int r = 0; r < M; r++)
                    memcpy(full + (size_t)r * Dmax + cur, outputs[i] + (size_t)r * d + cur, sizeof(float) * (d - cur));
                cur = d;
            }
        }
    }
    for (int i = 0; i < m->ns; i++) free(outputs[i]);
    if (tap == 100) {
        int64_t n = (int64_t)M * Dmax;
        int rc = 0;
        if (n > tap_cap) rc = fail("tap buffer too small");
        else { memcpy(tap_out, full, sizeof(float) * n); *tap_n = n; }
        free(full);
        return rc;
    }
    int Tp;
    float* dsd = simple_downsample(full, W(m, "encoder.downsample_output.bias"), 2, B, T50, Dmax, &Tp);
    free(full);
    if (m->ctc) ctc_head(m, dsd, B * Tp, enc_out);
    else linear(enc_out, m->J, dsd, Dmax, WT(m, m->J, Dmax, "joiner.encoder_proj.weight"), W(m, "joiner.encoder_proj.bias"), B * Tp, Dmax, m->J);
    free(dsd);
    return 0;
}

int k2o_offline_encoder(const k2o_model* m, const float* x, int B, int T, float* enc_out) {
    int64_t n;
    if (m->conformer) return conformer_forward(m, x, B, T, enc_out, -1, NULL, 0, &n);
    if (m->lstm) return lstm_forward(m, x, B, T, enc_out, -1, NULL, 0, &n);
    if (m->zip1) return zip1_offline_forward(m, x, B, T, enc_out, -1, NULL, 0, &n);
    return encoder_forward(m, x, B, T, enc_out, -1, NULL, 0, &n);
}
int64_t k2o_offline_encoder_tap(const k2o_model* m, const float* x, int B, int T, int tap, float* out, int64_t cap) {
    int64_t n = 0;
    int rc = m->conformer ? conformer_forward(m, x, B, T, NULL, tap, out, cap, &n)
             : m->lstm    ? lstm_forward(m, x, B, T, NULL, tap, out, cap, &n)
             : m->zip1    ? zip1_offline_forward(m, x, B, T, NULL, tap, out, cap, &n)
                          : encoder_forward(m, x, B, T, NULL, tap, out, cap, &n);
    return rc < 0 ? rc : n;
}

/* ------------------------------------------------------------------------- */
/* F5: stateless decoder (icefall decoder.py + OnnxDecoder):                 */
/*   emb(y.clamp(0)) * (y>=0) -> grouped Conv1d(k=ctx, groups=D/4, no bias)  */
/*   -> ReLU -> joiner.decoder_proj                                          */
/* Q9: the reference feeds id -1 at utterance start (OfflineRecognizer.cs:105)*/
/* ------------------------------------------------------------------------- */
int k2o_decoder(const k2o_model* m, const int64_t* y, int N, float* dec_out) {
    int DD = m->DD, ctx = m->ctx, V = m->V;
    const float* emb = W(m, "decoder.embedding.weight");
    const float* cw = W(m, "decoder.conv.weight"); /* [DD, cpg, ctx]; groups = DD / cpg */
    const int cpg = m->conv_cpg;
    const float* pwt = WT(m, m->J, DD, "joiner.decoder_proj.weight");
    const float* pb = W(m, "joiner.decoder_proj.bias");
    float* h = falloc((size_t)N * DD);
    for (int n = 0; n < N; n++) {
        for (int k = 0; k < ctx; k++)
            if (y[n * ctx + k] >= V) { free(h); return fail("decoder: token id %ld out of range", (long)y[n * ctx + k]); }
        for (int co = 0; co < DD; co++) {
            int g = co / cpg;
            float s = 0.f;
            for (int ci = 0; ci < cpg; ci++)
                for (int k = 0; k < ctx; k++) {
                    int64_t id = y[n * ctx + k];
                    float e = id >= 0 ? emb[(size_t)id * DD + g * cpg + ci] : 0.f;
                    s += cw[((size_t)co * cpg + ci) * ctx + k] * e;
                }
            h[(size_t)n * DD + co] = s > 0.f ? s : 0.f;
        }
    }
    linear(dec_out, m->J, h, DD, pwt, pb, N, DD, m->J);
    free(h);
    return 0;
}

/* F6: OnnxJoiner: output_linear(tanh(encoder_out + decoder_out)) */
int k2o_joiner(const k2o_model* m, const float* enc, const float* dec, int N, float* logits) {
    int J = m->J;
    float* a = falloc((size_t)N * J);
    for (size_t i = 0; i < (size_t)N * J; i++) a[i] = tanhf(enc[i] + dec[i]);
    linear(logits, m->V, a, J, WT(m, m->V, J, "joiner.output_linear.weight"), W(m, "joiner.output_linear.bias"), N, J, m->V);
    free(a);
    return 0;
}

/* OfflineRecognizer.cs:151-154 / :237-240:
 *   token_num = logits[j, token_num] > logits[j, k] ? token_num : k;   (k = 1..V-1)
 * ties and NaN comparisons move to the later index (Q6). */
int k2o_argmax_ref(const float* l, int V) {
    int tok = 0;
    for (int k = 1; k < V; k++) tok = l[tok] > l[k] ? tok : k;
    return tok;
}

static float top2_margin(const float* l, int V) {
    float a = -INFINITY, b = -INFINITY;
    for (int k = 0; k < V; k++) {
        if (l[k] > a) { b = a; a = l[k]; }
        else if (l[k] > b) b = l[k];
    }
    return a - b;
}

/* ------------------------------------------------------------------------- */
/* F7: ForwardBatchGreedySearch -- literal restatement of                    */
/* OfflineRecognizer.cs:189-303.                                             */
/*  - first decoder input is [-1, blank] for every stream (:202-208)         */
/*  - tokens[m] is seeded with 2*B blanks on the first frame (:250-258, Q10) */
/*  - a stream emits when y != blank(0) && y != unk(2) (:268, Q7)            */
/*  - when ANY stream emitted, the decoder is re-run for the WHOLE batch on   */
/*    the last ctx entries of every tokens[m] (:278-286).  For a stream that  */
/*    has not emitted yet those entries are the seeded blanks, so its context */
/*    silently changes from [-1,0] to [0,0] at the first emission of any      */
/*    other stream in the batch.                                              */
/* ------------------------------------------------------------------------- */
int k2o_greedy_batch(const k2o_model* m, const float* enc_out, int B, int Tp, int64_t* tokens, int32_t* timestamps,
                     int32_t* n_tokens, int max_tokens, float* margins) {
    int J = m->J, V = m->V, ctx = m->ctx;
    const int blank = 0, unk = 2;
    if (ctx != 2) return fail("greedy: context_size %d != 2 (reference hard-codes a 2-entry initial hyp)", ctx);
    int64_t* hyps = (int64_t*)malloc(sizeof(int64_t) * ctx * B);
    for (int i = 0; i < B; i++) { hyps[i * ctx] = -1; hyps[i * ctx + 1] = blank; }
    float* dec = falloc((size_t)B * J);
    if (k2o_decoder(m, hyps, B, dec)) { free(hyps); free(dec); return -1; }
    /* per-stream token lists with the reference's 2*B blank prefix */
    int cap = 2 * B + Tp + 1;
    int64_t* tk = (int64_t*)malloc(sizeof(int64_t) * (size_t)B * cap);
    int* ntk = (int*)calloc(B, sizeof(int));
    for (int i = 0; i < B; i++) n_tokens[i] = 0;
    float* cur = falloc((size_t)B * J);
    float* logits = falloc((size_t)B * V);
    int rc = 0;
    for (int t = 0; t < Tp && !rc; t++) {
        for (int b = 0; b < B; b++) memcpy(cur + (size_t)b * J, enc_out + ((size_t)b * Tp + t) * J, sizeof(float) * J);
        k2o_joiner(m, cur, dec, B, logits);
        int emitted = 0;
        for (int b = 0; b < B; b++) {
            int y = k2o_argmax_ref(logits + (size_t)b * V, V);
            if (margins) margins[(size_t)b * Tp + t] = top2_margin(logits + (size_t)b * V, V);
            if (ntk[b] == 0) {
                for (int i = 0; i < 2 * B; i++) tk[(size_t)b * cap + i] = blank;
                ntk[b] = 2 * B;
            }
            if (y != blank && y != unk) {
                tk[(size_t)b * cap + ntk[b]++] = y;
                if (n_tokens[b] >= max_tokens) { rc = fail("greedy: stream %d exceeds max_tokens %d", b, max_tokens); break; }
                tokens[(size_t)b * max_tokens + n_tokens[b]] = y;
                timestamps[(size_t)b * max_tokens + n_tokens[b]] = t;
                n_tokens[b]++;
                emitted = 1;
            }
        }
        if (emitted && !rc) {
            for (int b = 0; b < B; b++)
                for (int k = 0; k < ctx; k++) hyps[b * ctx + k] = tk[(size_t)b * cap + ntk[b] - ctx + k];
            rc = k2o_decoder(m, hyps, B, dec);
        }
    }
    free(hyps); free(dec); free(tk); free(ntk); free(cur); free(logits);
    return rc;
}

/* ForwardGreedySearch -- OfflineRecognizer.cs:93-187 (B=1, <=1 symbol per
 * frame via _max_sym_per_frame=1 :19,:129-134, max_sym_per_utt=1000 :122). */
int k2o_greedy_single(const k2o_model* m, const float* enc_out, int Tp, int64_t* tokens, int32_t* timestamps,
                      int32_t* n_tokens, int max_tokens, float* margins) {
    int J = m->J, V = m->V, ctx = m->ctx;
    const int blank = 0, unk = 2, max_sym_per_frame = 1, max_sym_per_utt = 1000;
    if (ctx != 2) return fail("greedy: context_size %d != 2", ctx);
    int cap = ctx + Tp + 1;
    int64_t* hyp = (int64_t*)malloc(sizeof(int64_t) * cap);
    int nh = 0;
    hyp[nh++] = -1;
    hyp[nh++] = blank;
    float* dec = falloc(J);
    float* logits = falloc(V);
    int rc = k2o_decoder(m, hyp, 1, dec);
    int t = 0, sym_per_frame = 0, sym_per_utt = 0;
    *n_tokens = 0;
    while (!rc && t < Tp && sym_per_utt < max_sym_per_utt) {
        if (sym_per_frame >= max_sym_per_frame) { sym_per_frame = 0; t += 1; continue; }
        k2o_joiner(m, enc_out + (size_t)t * J, dec, 1, logits);
        int y = k2o_argmax_ref(logits, V);
        if (margins) margins[t] = top2_margin(logits, V);
        if (y != blank && y != unk) {
            hyp[nh++] = y;
            if (*n_tokens >= max_tokens) { rc = fail("greedy: exceeds max_tokens %d", max_tokens); break; }
            tokens[*n_tokens] = y;
            timestamps[*n_tokens] = t;
            (*n_tokens)++;
            rc = k2o_decoder(m, hyp + nh - ctx, 1, dec);
            sym_per_utt += 1;
            sym_per_frame += 1;
        } else {
            sym_per_frame = 0;
            t += 1;
        }
    }
    free(hyp); free(dec); free(logits);
    return rc;
}

/* ForwardBatchGreedySearchCTC (OfflineRecognizer.cs:366-424) / ForwardGreedySearchCTC (:305-364) over log_probs
 * [B,T',V]:  y = Array.IndexOf(frame, frame.Max())  -- the FIRST index of the maximum (unlike the transducer loops);
 * emit when y != blank && y != prev_id (prev_id = -1 at the start of every call); timestamp = t + frame_offset[b];
 * num_trailing_blank[b] counts blanks since the last non-blank frame (accumulates onto its input value). */
int k2o_ctc_greedy(const float* log_probs, int B, int Tp, int V, const int32_t* frame_offsets, int64_t* tokens,
                   int32_t* timestamps, int32_t* n_tokens, int max_tokens, int32_t* num_trailing_blank) {
    const int blank = 0;
    for (int b = 0; b < B; b++) {
        int64_t prev = -1;
        int n = 0;
        for (int t = 0; t < Tp; t++) {
            const float* l = log_probs + ((size_t)b * Tp + t) * V;
            float mx = l[0];
            for (int v = 1; v < V; v++) mx = l[v] > mx ? l[v] : mx;   /* Enumerable.Max */
            int y = 0;
            while (y < V && !(l[y] == mx)) y++;                       /* Array.IndexOf: first match */
            if (y == V) y = 0;
            if (num_trailing_blank) num_trailing_blank[b] = (y == blank) ? num_trailing_blank[b] + 1 : 0;
            if (y != blank && y != prev) {
                if (n >= max_tokens) return fail("ctc greedy: stream %d exceeds max_tokens %d", b, max_tokens);
                tokens[(size_t)b * max_tokens + n] = y;
                timestamps[(size_t)b * max_tokens + n] = t + (frame_offsets ? frame_offsets[b] : 0);
                n++;
            }
            prev = y;
        }
        n_tokens[b] = n;
    }
    return 0;
}

/* GetResults (OfflineRecognizer.cs:85-91): EncoderProj (pad -> encoder) + batch greedy */
int k2o_offline_recognize_batch(const k2o_model* m, const float* const* feats, const int64_t* n_floats, int B,
                                int64_t* tokens, int32_t* timestamps, int32_t* n_tokens, int max_tokens) {
    int64_t L = k2o_pad_sequence(feats, n_floats, B, 19, NULL);
    float* x = falloc((size_t)B * L);
    k2o_pad_sequence(feats, n_floats, B, 19, x);
    /* OfflineProjOfTransducer.cs:59: T = padSequence.Length / FeatureDim / batchSize */
    int T = (int)(L / m->feat);
    int Tp = k2o_encoder_out_frames(m, T);
    if (Tp <= 0) { free(x); return fail("utterances too short: T=%d", T); }
    if (m->ctc) return fail("recognize_batch: CTC models decode through k2o_ctc_greedy");
    float* enc = falloc((size_t)B * Tp * m->J);
    int rc = k2o_offline_encoder(m, x, B, T, enc);
    free(x);
    if (!rc) rc = k2o_greedy_batch(m, enc, B, Tp, tokens, timestamps, n_tokens, max_tokens, NULL);
    free(enc);
    return rc;
}

/* streaming (OnlineRecognizer) path: same translation unit, shares the static helpers above */
#include "k2_oracle_beam.c"
#include "k2_oracle_online.c"
