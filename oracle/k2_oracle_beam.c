/*
 * k2_oracle_beam.c -- CPU restatement of modified beam search (included by k2_oracle.c; TEST
 * INFRASTRUCTURE ONLY, see k2_oracle.h).  PARITY UNPINNED, and more: the reference has NO beam search
 * at all -- OfflineRecognizer.cs:54-68 only knows "greedy_search" (SURVEY.md section 0).
 * BASELINE.json configs[2] asks for "modified-beam-search beam=4", so this file restates the published
 * icefall algorithm (beam_search.py modified_beam_search, batch form, temperature 1), per stream:
 *
 *   B = { ys = [blank]*context_size, log_prob = 0 }
 *   for t in 0..T'-1:
 *     A = list(B) in insertion order; B = {}
 *     for each hyp in A: log_probs[hyp] = log_softmax(joiner(enc[t], decoder(hyp.ys[-ctx:]))) + hyp.log_prob
 *     top `beam` of the flattened (hyp, token) scores                       (torch.topk)
 *     for each (hyp, token) in top order: new_ys = hyp.ys (+ token unless token in {blank, unk}),
 *        timestamp appended with t on a real token; B.add(new) where add() merges equal ys by
 *        logaddexp and keeps the first-inserted hyp's timestamps            (HypothesisList.add)
 *   result = argmax over B of log_prob / len(ys)   (len counts the ctx blanks; get_most_probable(length_norm=True))
 *
 * Tie-breaks the published code leaves to the library are fixed here: top-k is by (score desc, flat index
 * asc); the final argmax keeps the first maximum in insertion order (Python's max()).
 * x_lens = T' for every stream on this path (OfflineProjOfTransducer.cs:66-70), so the packed-sequence
 * bookkeeping of the batch form reduces to B independent streams.
 */
typedef struct {
    int64_t* ys;    /* [cap] incl. the ctx-blank prefix */
    int32_t* ts;    /* [cap] */
    int n, nts;
    float lp;
} beam_hyp;

static float logaddexp_f(float a, float b) {
    float mx = a > b ? a : b, d = -fabsf(a - b);
    if (isinf(mx) && mx < 0) return mx;
    return mx + log1pf(expf(d));
}

/* margins (optional): [Tp+1] -- per frame the gap between the beam-th and (beam+1)-th candidate score (INF if
 * fewer candidates), and in [Tp] the gap between the best and second-best length-normalised final scores.
 * trace (optional): [B][Tp][4*beam + 1] int32 words -- the per-frame tap the GPU search is localised against
 * (tests/parity.py assert_beam_match): the 2*beam best candidates of the frame in (score desc, flat index asc)
 * order, flat index = hypothesis slot * V + token (-1 where the frame has fewer candidates) in words [0, 2 beam),
 * their scores (float bits) in words [2 beam, 4 beam), and in the last word the number of hypotheses that survive
 * the frame (after HypothesisList.add's merges).  Two searches with equal histories up to frame t-1 hold the same
 * hypotheses in the same slots, so the first frame whose first `beam` flat indexes (or survivor count) differ is
 * the frame at which the searches part, and the scores say by what margin the oracle decided there. */
int k2o_modified_beam_search_trace(const k2o_model* m, const float* enc_out, int B, int Tp, int beam, int64_t* tokens,
                                   int32_t* timestamps, int32_t* n_tokens, int max_tokens, float* scores, float* margins,
                                   int32_t* trace) {
    const int J = m->J, V = m->V, ctx = m->ctx, blank = 0, unk = 2;
    if (beam < 1 || beam > 16) return fail("beam search: beam %d out of range [1,16]", beam);
    if (ctx != 2) return fail("beam search: context_size %d != 2", ctx);
    const int cap = ctx + Tp + 1;
    int rc = 0;
    for (int b = 0; b < B && !rc; b++) {
        beam_hyp* A = (beam_hyp*)calloc(beam, sizeof(beam_hyp));
        beam_hyp* N = (beam_hyp*)calloc(beam, sizeof(beam_hyp));
        for (int k = 0; k < beam; k++) {
            A[k].ys = (int64_t*)malloc(sizeof(int64_t) * cap); A[k].ts = (int32_t*)malloc(sizeof(int32_t) * cap);
            N[k].ys = (int64_t*)malloc(sizeof(int64_t) * cap); N[k].ts = (int32_t*)malloc(sizeof(int32_t) * cap);
        }
        int nA = 1;
        for (int k = 0; k < ctx; k++) A[0].ys[k] = blank;
        A[0].n = ctx; A[0].nts = 0; A[0].lp = 0.f;
        float* dec = falloc((size_t)beam * J);
        float* cur = falloc((size_t)beam * J);
        float* lg = falloc((size_t)beam * V);
        int64_t* yin = (int64_t*)malloc(sizeof(int64_t) * beam * ctx);
        float* mg = margins ? margins + (size_t)b * (Tp + 1) : NULL;
        for (int t = 0; t < Tp && !rc; t++) {
            for (int k = 0; k < nA; k++) {
                for (int c = 0; c < ctx; c++) yin[k * ctx + c] = A[k].ys[A[k].n - ctx + c];
                memcpy(cur + (size_t)k * J, enc_out + ((size_t)b * Tp + t) * J, sizeof(float) * J);
            }
            if ((rc = k2o_decoder(m, yin, nA, dec))) break;
            k2o_joiner(m, cur, dec, nA, lg);
            /* log_softmax (float32: x - max - log(sum(exp(x - max)))) + hyp log_prob */
            for (int k = 0; k < nA; k++) {
                float* l = lg + (size_t)k * V;
                float mx = l[0];
                for (int v = 1; v < V; v++) mx = l[v] > mx ? l[v] : mx;
                float s = 0.f;
                for (int v = 0; v < V; v++) s += expf(l[v] - mx);
                float lse = logf(s);
                for (int v = 0; v < V; v++) l[v] = (l[v] - mx - lse) + A[k].lp;
            }
            /* top `beam` of nA*V by (value desc, flat index asc), plus the runner-up for the margin */
            int nc = nA * V, want = beam < nc ? beam : nc;
            const int ext = trace ? 2 * beam : want + 1;   /* how many candidates to rank */
            int top[33];
            float topv[33];
            int nt = 0;
            for (int r = 0; r < ext && r < nc; r++) {
                int bi = -1;
                float bv = -INFINITY;
                for (int i = 0; i < nc; i++) {
                    int taken = 0;
                    for (int q = 0; q < nt; q++) taken |= (top[q] == i);
                    if (taken) continue;
                    if (bi < 0 || lg[i] > bv) { bi = i; bv = lg[i]; }
                }
                top[nt] = bi; topv[nt] = bv; nt++;
            }
            if (mg) mg[t] = nt > want ? topv[want - 1] - topv[want] : INFINITY;
            int nN = 0;
            for (int r = 0; r < want; r++) {
                int hi = top[r] / V, tok = top[r] % V;
                const beam_hyp* h = &A[hi];
                int real = (tok != blank && tok != unk);
                int nn = h->n + real;
                /* HypothesisList.add: merge when the token sequence already exists */
                int dup = -1;
                for (int q = 0; q < nN && dup < 0; q++) {
                    if (N[q].n != nn) continue;
                    int same = memcmp(N[q].ys, h->ys, sizeof(int64_t) * h->n) == 0;
                    if (same && real) same = N[q].ys[nn - 1] == tok;
                    if (same) dup = q;
                }
                if (dup >= 0) { N[dup].lp = logaddexp_f(N[dup].lp, topv[r]); continue; }
                beam_hyp* d = &N[nN++];
                memcpy(d->ys, h->ys, sizeof(int64_t) * h->n);
                memcpy(d->ts, h->ts, sizeof(int32_t) * h->nts);
                d->n = h->n; d->nts = h->nts;
                if (real) { d->ys[d->n++] = tok; d->ts[d->nts++] = t; }
                d->lp = topv[r];
            }
            if (trace) {
                int32_t* tr = trace + ((size_t)b * Tp + t) * (4 * beam + 1);
                for (int r = 0; r < 2 * beam; r++) {
                    tr[r] = r < nt ? top[r] : -1;
                    float v = r < nt ? topv[r] : -INFINITY;
                    memcpy(&tr[2 * beam + r], &v, sizeof(float));
                }
                tr[4 * beam] = nN;
            }
            beam_hyp* tmp = A; A = N; N = tmp;
            nA = nN;
        }
        if (!rc) {
            int best = 0;
            float bs = A[0].lp / (float)A[0].n, second = -INFINITY;
            for (int k = 1; k < nA; k++) {
                float s = A[k].lp / (float)A[k].n;
                if (s > bs) { second = bs; bs = s; best = k; }
                else if (s > second) second = s;
            }
            if (mg) mg[Tp] = nA > 1 ? bs - second : INFINITY;
            int n = A[best].n - ctx;
            if (n > max_tokens) rc = fail("beam search: stream %d has %d tokens > max_tokens %d", b, n, max_tokens);
            else {
                for (int i = 0; i < n; i++) {
                    tokens[(size_t)b * max_tokens + i] = A[best].ys[ctx + i];
                    timestamps[(size_t)b * max_tokens + i] = A[best].ts[i];
                }
                n_tokens[b] = n;
                if (scores) scores[b] = A[best].lp;
            }
        }
        for (int k = 0; k < beam; k++) { free(A[k].ys); free(A[k].ts); free(N[k].ys); free(N[k].ts); }
        free(A); free(N); free(dec); free(cur); free(lg); free(yin);
    }
    return rc;
}

int k2o_modified_beam_search(const k2o_model* m, const float* enc_out, int B, int Tp, int beam, int64_t* tokens,
                             int32_t* timestamps, int32_t* n_tokens, int max_tokens, float* scores, float* margins) {
    return k2o_modified_beam_search_trace(m, enc_out, B, Tp, beam, tokens, timestamps, n_tokens, max_tokens, scores, margins, NULL);
}
