/*
 * wtext_core.h — the sequential side of `wtext` (one of the two consumers of the CIGAR column wtzmo writes; wtext.c:129-297, 377-540):
 * options, reads and retained regions, the overlap lines, the clipping of an overlap's CIGAR to the retained regions of its two reads, the
 * re-scoring of what is left, and the assembly of the extended record.  The two end extensions of an overlap - kswx_extend_align, i.e.
 * kswx_extend_align_shift_core (kswx.h:469-481, 101-232), the routine behind wtzmo's K-sw3 kernel - are NOT here: the includer defines
 * wx_extend_jobs() (device: wtext_main.c through wtz_extend_batch; CPU oracle: oracle/wtext_oracle.c).
 *
 * Why the device can take batches: an overlap's work is independent of every other overlap's (the reference's workers share nothing but the
 * read set); inside one overlap the right extension starts from the score the left extension returned (wtext.c:250-268).  So a block of
 * overlaps is processed in three sweeps - clip + re-score on the host, all left extensions as one device batch, all right extensions as
 * another - and written in input order, which is the order of `wtext -t 1` (with more threads the reference writes batches of 100 lines in
 * the order its workers finish).
 *
 * Kept because they are in the reference's output: `-S` is in the usage text but not in the getopt string (wtext.c:424: it prints the usage);
 * `-P/-p` deal BATCHES OF 100 INPUT LINES, skipped lines included (wtext.c:489-494); a record is printed iff its alignment length is > 0
 * (wtext.c:329), which drops every overlap whose CIGAR does not survive the clipping; the three CIGAR pieces are joined as TEXT, so equal
 * operations on either side of a joint stay separate ("12M" "30M").
 */
#ifndef WTEXT_CORE_H
#define WTEXT_CORE_H

#include <getopt.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include "wtz_host.h"

#define WX_DIE() do { fflush(NULL); _exit(1); } while(0)      /* the device context may be under construction on a helper thread: never exit() under it */

typedef struct { char **a; int n, cap; } wx_strlist_t;
static void wx_sl_push(wx_strlist_t *l, char *s){ if(l->n == l->cap){ l->cap = l->cap ? l->cap * 2 : 4; l->a = (char**)hx_realloc(l->a, sizeof(char*) * (size_t)l->cap); } l->a[l->n++] = s; }

typedef struct {
	int ncpu, n_job, i_job, W, M, X, O, E, T, max_ext, overwrite;
	wx_strlist_t pbs, ovls, obts, cycs; char *output;
	int gpu; uint64_t pool_gb; uint32_t block;      /* ours: device, scratch, overlaps per device block */
} wx_opt_t;

typedef struct { int32_t score, tb, te, qb, qe, aln, mat, mis, ins, del; } wx_aln_t;      /* kswx_t */

/* one end extension: `len` bases of read `rd` (strand `rev`) walking `strand` from view position `from` */
typedef struct { uint32_t q_read, q_rev, t_read, t_rev; int32_t q_from, t_from, strand, q_len, t_len, init_score; } wx_job_t;
typedef struct { wx_aln_t x; uint64_t cig_off; uint32_t cig_len; } wx_jobres_t;       /* operations (len << 4 | op) in *cigar_pool, first operation first */

typedef struct {
	uint32_t pb1, pb2; uint8_t dir1, dir2, alive, need_l, need_r;
	int32_t tb, te, qb, qe;                 /* whole-read coordinates of the input line (previous retained region undone) */
	char *cigar_in;
	uint32_t *core; uint32_t ncore;         /* operations that survive the clipping */
	wx_aln_t x0;
	int32_t clplen[2], dx[2];
	uint32_t jl, jr;                        /* job index of the left / right extension */
} wx_hit_t;

typedef struct {
	wx_opt_t O;
	hx_store_t st; uint32_t n_pb; hx_names_t names;
	uint32_t *pblen, *prev_off, *prev_len, *clp_off, *clp_len;
	wx_hit_t *hits; size_t nhit, caphit;
	wx_job_t *jobs; wx_jobres_t *res; size_t njob, capjob;
	uint32_t *cigar_pool; uint64_t ncig, capcig;
	FILE *out; unsigned long long n_in, n_out, n_ext;
	double t_read, t_clip, t_ext, t_write;      /* host seconds: reading + splitting the lines, clip + re-score, the two extension sweeps, record assembly */
	void *backend;
} wx_t;

/* defined by the includer: run jobs[0..n), fill res[0..n) (operations appended to W->cigar_pool through wx_cigar_space) */
static void wx_extend_jobs(wx_t *W, const wx_job_t *jobs, size_t n, wx_jobres_t *res);

static uint32_t *wx_cigar_space(wx_t *W, uint64_t n){
	if(W->ncig + n + 1 > W->capcig){ uint64_t c = W->capcig ? W->capcig : (1u << 20); while(c < W->ncig + n + 1) c += c / 2; W->cigar_pool = (uint32_t*)hx_realloc(W->cigar_pool, 4 * c); W->capcig = c; }
	uint32_t *p = W->cigar_pool + W->ncig; W->ncig += n; return p;
}
static const char *wx_date(void){ static char buf[64]; time_t t = time(NULL); struct tm tmv; localtime_r(&t, &tmv); strftime(buf, sizeof buf, "%a %b %e %H:%M:%S %Y", &tmv); return buf; }

static int wx_usage(void){      /* wtext.c:342-375 prints to stdout and returns 1; letters and defaults are the contract, the wording is ours */
	fputs(
	"WTEXT (MI355X tool set): overlaps clipped to the retained regions of their reads and extended to the region ends (kswx_extend_align on gfx950)\n"
	"Usage: wtext [options]      (* = required, + = may be given several times)\n"
	" inputs   -i <reads fa/fq[.gz]> *+   -j <overlap file with the CIGAR column (17 columns)> *+\n"
	"          -b <name offset length: retained region, e.g. from wtobt> +   -B <the same for the region the overlaps' coordinates refer to, e.g. from wtcyc> +\n"
	" output   -o <extended overlaps, - = stdout> *   -f overwrite\n"
	" jobs     -t <int> accepted (the output is that of -t 1)   -P <int> -p <int> this job of that many: batches of 100 input lines are dealt round-robin\n"
	" scores   -W <int> band [800]   -M 2 -X -5 -O -3 -E -1   -T <int> end clipping [-100]\n"
	" this build --gpu <id>   --pool-gb <n> [16]   --block <overlaps per device block> [65536]\n",
	stdout);
	return 1;
}
static int wx_parse_args(wx_opt_t *o, int argc, char **argv){
	memset(o, 0, sizeof *o);
	o->ncpu = 1; o->n_job = 1; o->i_job = 0; o->W = 800; o->M = 2; o->X = -5; o->O = -3; o->E = -1; o->T = -100; o->max_ext = 400;      /* wtext.c:397-404 */
	o->block = 65536;
	static const struct option lopts[] = { {"gpu", 1, 0, 1001}, {"pool-gb", 1, 0, 1002}, {"block", 1, 0, 1003}, {0, 0, 0, 0} };
	int c; optind = 1;
	while((c = getopt_long(argc, argv, "hft:P:p:i:B:b:j:o:W:M:X:O:E:T:s:m:", lopts, NULL)) != -1){      /* wtext.c:406: no 'S' */
		switch(c){
			case 'h': return 1;
			case 'f': o->overwrite = 1; break;
			case 't': o->ncpu = atoi(optarg); break;
			case 'P': o->n_job = atoi(optarg); break;
			case 'p': o->i_job = atoi(optarg); break;
			case 'i': wx_sl_push(&o->pbs, optarg); break;
			case 'B': wx_sl_push(&o->cycs, optarg); break;
			case 'b': wx_sl_push(&o->obts, optarg); break;
			case 'j': wx_sl_push(&o->ovls, optarg); break;
			case 'o': o->output = optarg; break;
			case 'W': o->W = atoi(optarg); break;
			case 'M': o->M = atoi(optarg); break;
			case 'X': o->X = atoi(optarg); break;
			case 'O': o->O = atoi(optarg); break;
			case 'E': o->E = atoi(optarg); break;
			case 'T': o->T = atoi(optarg); break;
			case 's': case 'm': break;                                 /* parsed, never used (wtext.c:424-425; the filters are commented out of the usage) */
			case 1001: o->gpu = atoi(optarg); break;
			case 1002: o->pool_gb = (uint64_t)atoll(optarg); break;
			case 1003: o->block = (uint32_t)atoi(optarg); if(o->block < 1) o->block = 1; break;
			default: return 1;
		}
	}
	if(o->pbs.n == 0 || o->ovls.n == 0 || o->output == NULL) return 1;
	if(!o->overwrite && strcmp(o->output, "-") && access(o->output, F_OK) == 0){ fprintf(stderr, "File exists! '%s'\n\n", o->output); return 1; }
	return 0;
}

/* split_string (string.h:251-274): tab-separated, empty fields vanish */
static int wx_split_tabs(char *line, char **col, int maxcol){
	int n = 0; char *p = line, *s = line;
	for(;; p++){
		if(*p == '\t' || *p == 0){
			const int end = (*p == 0);
			if(p > s){ if(n < maxcol) col[n] = s; n++; *p = 0; }
			s = p + 1;
			if(end) break;
		}
	}
	return n;
}

static void wx_load_inputs(wx_t *W){
	wx_opt_t *o = &W->O;
	hx_reader_t *fr = hx_reader_open(o->pbs.a, o->pbs.n);
	if(!fr){ fprintf(stderr, " -- Cannot open %s --\n", o->pbs.a[0]); WX_DIE(); }
	fprintf(stderr, "[%s] loading the reads\n", wx_date());
	hx_str_t name = {0, 0, 0}, seq = {0, 0, 0};
	while(hx_reader_seq(fr, &name, &seq)) hx_store_add(&W->st, name.s ? name.s : "", name.n, seq.s ? seq.s : "", seq.n);      /* file order = read id (wtext.c:94-110) */
	hx_reader_close(fr);
	W->n_pb = W->st.n_all;
	fprintf(stderr, "[%s] %u reads\n", wx_date(), W->n_pb);
	hx_names_build(&W->names, W->st.reads, W->n_pb);
	const size_t n = (size_t)W->n_pb + 1;
	W->pblen = (uint32_t*)hx_realloc(NULL, 4 * n); W->prev_off = (uint32_t*)calloc(n, 4); W->prev_len = (uint32_t*)hx_realloc(NULL, 4 * n);
	W->clp_off = (uint32_t*)calloc(n, 4); W->clp_len = (uint32_t*)hx_realloc(NULL, 4 * n);
	for(uint32_t i = 0; i < W->n_pb; i++) W->pblen[i] = W->prev_len[i] = W->clp_len[i] = W->st.reads[i].len;
	char *col[20];
	for(int pass = 0; pass < 2; pass++){         /* -B (wtext.c:441-450: no comment lines), then -b (wtext.c:451-461) */
		wx_strlist_t *l = pass ? &o->obts : &o->cycs;
		if(l->n == 0) continue;
		if((fr = hx_reader_open(l->a, l->n)) == NULL) WX_DIE();
		while(hx_reader_line(fr) != -1){
			const int first_hash = (fr->line[0] == '#');
			const int nc = wx_split_tabs(fr->line, col, 20);
			if(pass && first_hash) continue;
			if(nc < 3) continue;
			const uint32_t id = hx_names_get(&W->names, col[0]);
			if(id == 0xFFFFFFFFu) continue;
			const int coff = atoi(col[1]), clen = atoi(col[2]);
			if(coff < 0 || coff + clen > (int)W->pblen[id]) continue;
			if(pass){ W->clp_off[id] = (uint32_t)coff; W->clp_len[id] = (uint32_t)clen; } else { W->prev_off[id] = (uint32_t)coff; W->prev_len[id] = (uint32_t)clen; }
		}
		hx_reader_close(fr);
	}
}

/* base of read `id` at position p of its (whole-read) view, reverse-complemented when rev: bitseq_basebank / revbitseq_basebank */
static inline unsigned wx_base(const wx_t *W, uint32_t id, int rev, int64_t p){
	const uint64_t off = W->st.reads[id].off; const uint32_t len = W->pblen[id];
	const uint64_t x = rev ? off + len - 1 - (uint64_t)p : off + (uint64_t)p;
	const unsigned b = (unsigned)(W->st.bits[x >> 5] >> (((~x) & 31u) << 1)) & 3u;
	return rev ? 3u - b : b;
}

/* kswx_string2cigar (kswx.h:1122-1148): equal neighbours merge, anything that is not M / I / D is operation 3 */
static uint32_t wx_parse_cigar(const char *s, uint32_t **out, uint32_t *cap){
	uint32_t n = 0, len = 0;
	for(const char *p = s; *p; p++){
		if(*p >= '0' && *p <= '9'){ len = len * 10 + (uint32_t)(*p - '0'); continue; }
		const uint32_t op = *p == 'M' ? 0u : (*p == 'I' ? 1u : (*p == 'D' ? 2u : 3u));
		if(n && ((*out)[n - 1] & 0xFu) == op) (*out)[n - 1] += len << 4;
		else { if(n == *cap){ *cap = *cap ? *cap * 2 : 256; *out = (uint32_t*)hx_realloc(*out, 4 * (size_t)*cap); } (*out)[n++] = (len << 4) | op; }
		len = 0;
	}
	return n;
}

/* ---- an overlap against the retained regions of its two reads (what wtext.c:129-244 computes), in three steps of its own:
 *   1. how many bases of each read the alignment must give up at its front and at its back (it may poke out of a region at either end),
 *   2. wx_trim_end eats operations from that end until both reads have given up enough (a partly eaten match run is shortened in place),
 *   3. wx_rescore walks what is left once, base by base, for the record's counts and score.
 * Kept because the output shows it: gap operations at an end are eaten before the first match run is even looked at, whether or not anything had to be given up
 * there; a match run gives up max(lack of read 1, lack of read 2) bases of BOTH reads; an end that runs out of operations, or two ends that meet, drop the record. ---- */
typedef struct { int t, q; } wx_both_t;       /* one number per read: t = read 1 (the overlap's target columns), q = read 2 */

/* returns the operations used up entirely (>= 0), *gave = the bases each read gave up; -1 when the list ends before both reads gave up `owe` */
static int wx_trim_end(uint32_t *cg, uint32_t nc, int from_back, wx_both_t owe, wx_both_t *gave){
	wx_both_t got = {0, 0}; uint32_t used = 0;
	for(; used < nc; used++){
		uint32_t *w = &cg[from_back ? nc - 1 - used : used];
		const uint32_t op = *w & 0xFu; const int run = (int)(*w >> 4);
		if(op == 1){ got.q += run; continue; }                    /* an insertion: bases of read 2 only */
		if(op == 2){ got.t += run; continue; }                    /* a deletion: bases of read 1 only */
		if(got.t >= owe.t && got.q >= owe.q) break;               /* a match run, and nothing is owed any more: the kept part starts here */
		const int lack_t = owe.t - got.t, lack_q = owe.q - got.q;
		int take = lack_t > lack_q ? lack_t : lack_q; if(take > run) take = run;
		got.t += take; got.q += take;
		if(take < run){ *w = ((uint32_t)(run - take) << 4) | op; break; }
	}
	*gave = got;
	return (got.t < owe.t || got.q < owe.q) ? -1 : (int)used;
}

/* counts and score of the kept operations cg[0, n), read 1 from base at1, read 2 from base at2 (positions on the strands the overlap was reported on) */
static void wx_rescore(const wx_t *W, const wx_hit_t *h, const uint32_t *cg, uint32_t n, int64_t at1, int64_t at2, wx_aln_t *x){
	const wx_opt_t *o = &W->O;
	for(uint32_t i = 0; i < n; i++){
		const int op = (int)(cg[i] & 0xFu), run = (int)(cg[i] >> 4);
		x->aln += run;
		if(op == 1){ x->ins += run; at2 += run; x->score += o->O + o->E * run; }
		else if(op == 2){ x->del += run; at1 += run; x->score += o->O + o->E * run; }
		else {
			for(int j = 0; j < run; j++){ if(wx_base(W, h->pb1, h->dir1, at1 + j) == wx_base(W, h->pb2, h->dir2, at2 + j)) x->mat++; else x->mis++; }
			at1 += run; at2 += run;
		}
	}
	x->score += x->mat * o->M; x->score += x->mis * o->X;
}

/* alive = 0 afterwards: nothing of the overlap is left inside the regions (the record is dropped, wtext.c:329) */
static void wx_clip_hit(wx_t *W, wx_hit_t *h, uint32_t **tmp, uint32_t *captmp){
	const uint32_t nc = wx_parse_cigar(h->cigar_in, tmp, captmp);
	uint32_t *cg = *tmp;
	free(h->cigar_in); h->cigar_in = NULL;
	memset(&h->x0, 0, sizeof h->x0); h->alive = 0; h->need_l = h->need_r = 0; h->core = NULL; h->ncore = 0;
	const uint32_t rd[2] = { h->pb1, h->pb2 }; const int rev[2] = { h->dir1, h->dir2 };
	const int from[2] = { h->tb, h->qb }, upto[2] = { h->te, h->qe };
	int before[2], after[2];              /* bases of the read outside its region, in front of / behind it on the overlap's strand */
	for(int k = 0; k < 2; k++){
		const int L = (int)W->pblen[rd[k]], off = (int)W->clp_off[rd[k]], keep = (int)W->clp_len[rd[k]];
		h->clplen[k] = keep;
		before[k] = rev[k] ? L - off - keep : off;
		after[k] = rev[k] ? off : L - off - keep;
		h->dx[k] = before[k];
	}
	const wx_both_t owe_front = { before[0] > from[0] ? before[0] - from[0] : 0, before[1] > from[1] ? before[1] - from[1] : 0 };
	const int tail1 = (int)W->pblen[rd[0]] - upto[0], tail2 = (int)W->pblen[rd[1]] - upto[1];
	const wx_both_t owe_back = { after[0] > tail1 ? after[0] - tail1 : 0, after[1] > tail2 ? after[1] - tail2 : 0 };
	wx_both_t cut_front, cut_back;
	const int nf = wx_trim_end(cg, nc, 0, owe_front, &cut_front);
	if(nf < 0) return;
	const int nb = wx_trim_end(cg, nc, 1, owe_back, &cut_back);
	if(nb < 0 || nf + nb >= (int)nc) return;
	wx_aln_t x0; memset(&x0, 0, sizeof x0);
	x0.tb = from[0] + cut_front.t - before[0]; x0.qb = from[1] + cut_front.q - before[1];      /* region coordinates */
	x0.te = upto[0] - cut_back.t - before[0];  x0.qe = upto[1] - cut_back.q - before[1];
	h->ncore = nc - (uint32_t)nf - (uint32_t)nb;
	wx_rescore(W, h, cg + nf, h->ncore, (int64_t)before[0] + x0.tb, (int64_t)before[1] + x0.qb, &x0);
	h->core = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)h->ncore + 1)); memcpy(h->core, cg + nf, 4 * (size_t)h->ncore);
	h->x0 = x0; h->alive = 1;
	h->need_l = (x0.qb <= W->O.max_ext || x0.tb <= W->O.max_ext);                               /* wtext.c:247 */
}

static void wx_job_push(wx_t *W, const wx_job_t *j){
	if(W->njob == W->capjob){ W->capjob = W->capjob ? W->capjob * 2 : 4096; W->jobs = (wx_job_t*)hx_realloc(W->jobs, sizeof(wx_job_t) * W->capjob); W->res = (wx_jobres_t*)hx_realloc(W->res, sizeof(wx_jobres_t) * W->capjob); }
	W->jobs[W->njob++] = *j;
}
static inline size_t wx_put_int(char *o, long long v){ char t[24]; int n = 0; unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v; size_t k = 0; if(v < 0) o[k++] = '-'; do { t[n++] = (char)('0' + u % 10); u /= 10; } while(u); while(n) o[k++] = t[--n]; return k; }
/* kswx_cigar2string (kswx.h:1093-1120): zero-length operations vanish; an operation beyond M / I / D ends the program like the reference */
static size_t wx_cigar_text(char *o, const uint32_t *cg, uint32_t n, int reversed){
	size_t k = 0;
	for(uint32_t i = 0; i < n; i++){
		const uint32_t w = cg[reversed ? n - 1 - i : i], op = w & 0xFu, len = w >> 4;
		if(len == 0) continue;
		if(op > 2){ fprintf(stderr, " -- an operation other than M, I, D in a CIGAR --\n"); WX_DIE(); }
		k += wx_put_int(o + k, (long long)len); o[k++] = "MID"[op];
	}
	return k;
}

static double wx_now(void){ struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
/* the clipping of a hit reads the shared read set and writes its own hit: `-t` threads take contiguous slices (nothing observable depends on it) */
typedef struct { wx_t *W; size_t lo, hi; pthread_t th; int started; } wx_clipjob_t;
static void *wx_clip_main(void *arg){
	wx_clipjob_t *j = (wx_clipjob_t*)arg;
	uint32_t *tmp = NULL, captmp = 0;
	for(size_t i = j->lo; i < j->hi; i++) wx_clip_hit(j->W, &j->W->hits[i], &tmp, &captmp);
	free(tmp);
	return NULL;
}
static void wx_clip_all(wx_t *W){
	int nt = W->O.ncpu; if(nt > 64) nt = 64;
	if(nt < 2 || W->nhit < 256){ wx_clipjob_t j; memset(&j, 0, sizeof j); j.W = W; j.hi = W->nhit; wx_clip_main(&j); return; }
	wx_clipjob_t *js = (wx_clipjob_t*)hx_realloc(NULL, sizeof(wx_clipjob_t) * (size_t)nt);
	for(int t = 0; t < nt; t++){
		js[t].W = W; js[t].lo = W->nhit * (size_t)t / (size_t)nt; js[t].hi = W->nhit * (size_t)(t + 1) / (size_t)nt;
		js[t].started = (pthread_create(&js[t].th, NULL, wx_clip_main, &js[t]) == 0);
		if(!js[t].started) wx_clip_main(&js[t]);
	}
	for(int t = 0; t < nt; t++) if(js[t].started) pthread_join(js[t].th, NULL);
	free(js);
}

/* output_alignments_wtext, wtext.c:323-340: the records of a slice of hits as text; `-t` threads each take a contiguous slice, the slices are written in order */
typedef struct { wx_t *W; size_t lo, hi; char *buf; size_t n, cap; unsigned long long n_out; pthread_t th; int started; } wx_fmtjob_t;
static void *wx_fmt_main(void *arg){
	wx_fmtjob_t *f = (wx_fmtjob_t*)arg; wx_t *W = f->W;
	const hx_read_t *R = W->st.reads;
	for(size_t i = f->lo; i < f->hi; i++){
		wx_hit_t *h = &W->hits[i];
		if(h->alive && h->x0.aln > 0){
			const uint32_t nl = h->need_l ? h->jl : 0, nr = h->need_r ? h->jr : 0;
			const size_t need = strlen(R[h->pb1].name) + strlen(R[h->pb2].name) + 256 + 12 * ((size_t)nl + h->ncore + nr);
			if(f->n + need > f->cap){ f->cap = (f->n + need) * 2; f->buf = (char*)hx_realloc(f->buf, f->cap); }
			char *buf = f->buf + f->n; size_t k = 0;
			k += (size_t)sprintf(buf + k, "%s\t%c\t%d\t%d\t%d", R[h->pb1].name, "+-"[h->dir1], (int)W->clp_len[h->pb1], h->x0.tb, h->x0.te);
			k += (size_t)sprintf(buf + k, "\t%s\t%c\t%d\t%d\t%d", R[h->pb2].name, "+-"[h->dir2], (int)W->clp_len[h->pb2], h->x0.qb, h->x0.qe);
			k += (size_t)sprintf(buf + k, "\t%d\t%0.3f\t%d\t%d\t%d\t%d\t", h->x0.score, 1.0 * h->x0.mat / h->x0.aln, h->x0.mat, h->x0.mis, h->x0.ins, h->x0.del);
			if(nl) k += wx_cigar_text(buf + k, (const uint32_t*)h->cigar_in, nl, 1);
			k += wx_cigar_text(buf + k, h->core, h->ncore, 0);
			if(nr) k += wx_cigar_text(buf + k, h->core + h->ncore, nr, 0);
			buf[k++] = '\n';
			f->n += k; f->n_out++;
		}
		free(h->cigar_in); free(h->core); h->cigar_in = NULL; h->core = NULL;
	}
	return NULL;
}
static void wx_write_records(wx_t *W){
	const double t0 = wx_now();
	int nt = W->O.ncpu; if(nt > 64) nt = 64; if(nt < 1 || W->nhit < 256) nt = 1;
	wx_fmtjob_t *fs = (wx_fmtjob_t*)calloc((size_t)nt, sizeof(wx_fmtjob_t));
	for(int t = 0; t < nt; t++){
		fs[t].W = W; fs[t].lo = W->nhit * (size_t)t / (size_t)nt; fs[t].hi = W->nhit * (size_t)(t + 1) / (size_t)nt;
		if(nt > 1) fs[t].started = (pthread_create(&fs[t].th, NULL, wx_fmt_main, &fs[t]) == 0);
		if(!fs[t].started) wx_fmt_main(&fs[t]);
	}
	for(int t = 0; t < nt; t++){
		if(fs[t].started) pthread_join(fs[t].th, NULL);
		if(fs[t].n) fwrite(fs[t].buf, 1, fs[t].n, W->out);
		W->n_out += fs[t].n_out; free(fs[t].buf);
	}
	free(fs);
	W->nhit = 0;
	W->t_write += wx_now() - t0;
}

/* the hits of the block: clip, left extensions, right extensions, records in input order */
static void wx_process_block(wx_t *W){
	const wx_opt_t *o = &W->O;
	double t0 = wx_now();
	wx_clip_all(W);
	W->t_clip += wx_now() - t0; t0 = wx_now();
	for(int side = 0; side < 2; side++){
		W->njob = 0; W->ncig = 0;
		for(size_t i = 0; i < W->nhit; i++){
			wx_hit_t *h = &W->hits[i];
			if(!h->alive) continue;
			if(side == 1) h->need_r = (h->clplen[1] - h->x0.qe <= o->max_ext || h->clplen[0] - h->x0.te <= o->max_ext);      /* wtext.c:265, with the coordinates the left extension left */
			if(side == 0 ? !h->need_l : !h->need_r) continue;
			wx_job_t j; memset(&j, 0, sizeof j);
			j.q_read = h->pb2; j.q_rev = h->dir2; j.t_read = h->pb1; j.t_rev = h->dir1; j.init_score = h->x0.score;
			if(side == 0){ j.strand = -1; j.q_len = h->x0.qb; j.t_len = h->x0.tb; j.q_from = h->dx[1] + h->x0.qb - 1; j.t_from = h->dx[0] + h->x0.tb - 1; h->jl = (uint32_t)W->njob; }
			else { j.strand = 1; j.q_len = h->clplen[1] - h->x0.qe; j.t_len = h->clplen[0] - h->x0.te; j.q_from = h->dx[1] + h->x0.qe; j.t_from = h->dx[0] + h->x0.te; h->jr = (uint32_t)W->njob; }
			wx_job_push(W, &j);
		}
		if(W->njob) wx_extend_jobs(W, W->jobs, W->njob, W->res);
		W->n_ext += W->njob;
		for(size_t i = 0; i < W->nhit; i++){
			wx_hit_t *h = &W->hits[i];
			if(!h->alive || (side == 0 ? !h->need_l : !h->need_r)) continue;
			const wx_jobres_t *r = &W->res[side == 0 ? h->jl : h->jr];
			h->x0.score = r->x.score; h->x0.aln += r->x.aln; h->x0.mat += r->x.mat; h->x0.mis += r->x.mis; h->x0.ins += r->x.ins; h->x0.del += r->x.del;
			if(side == 0){ h->x0.qb -= r->x.qe; h->x0.tb -= r->x.te; } else { h->x0.qe += r->x.qe; h->x0.te += r->x.te; }
			/* the piece's operations move to the hit (the pool is reused by the next sweep): the left piece into cigar_in (free since the clipping), the right piece behind the core */
			if(side == 0){ uint32_t *keep = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)r->cig_len + 1)); memcpy(keep, W->cigar_pool + r->cig_off, 4 * (size_t)r->cig_len); h->jl = r->cig_len; h->cigar_in = (char*)keep; }
			else { h->jr = r->cig_len; h->core = (uint32_t*)hx_realloc(h->core, 4 * ((size_t)h->ncore + r->cig_len + 1)); memcpy(h->core + h->ncore, W->cigar_pool + r->cig_off, 4 * (size_t)r->cig_len); }
		}
	}
	W->t_ext += wx_now() - t0; t0 = wx_now();
	wx_write_records(W);
}
static void *wx_block_main(void *arg){ wx_process_block((wx_t*)arg); return NULL; }

static int wx_run(wx_t *W){
	wx_opt_t *o = &W->O;
	W->out = strcmp(o->output, "-") ? fopen(o->output, "w") : stdout;
	if(!W->out){ fprintf(stderr, "Cannot open %s for write\n", o->output); return 1; }
	setvbuf(W->out, NULL, _IOFBF, 4u << 20);
	hx_reader_t *fr = hx_reader_open(o->ovls.a, o->ovls.n);
	if(!fr){ fprintf(stderr, " -- Cannot open %s --\n", o->ovls.a[0]); WX_DIE(); }
	fprintf(stderr, "[%s] extending the overlaps\n", wx_date());
	char **col = (char**)hx_realloc(NULL, sizeof(char*) * 24);
	unsigned long long nb = 0; int eof = 0;
	double t_rd = wx_now();
	/* the lines of the next block are read and split while the previous block is clipped, extended and written (a worker thread; `-t 1` keeps one thread) */
	wx_hit_t *rh = NULL; size_t rn = 0, rcap = 0;
	pthread_t worker; int worker_on = 0;
	while(!eof){
		const int mine = ((int)(nb++ % (unsigned long long)(o->n_job > 0 ? o->n_job : 1)) == o->i_job);       /* wtext.c:489: decided per batch of 100 LINES */
		for(int i = 0; i < 100; i++){
			if(hx_reader_line(fr) == -1){ eof = 1; break; }
			if(!mine) continue;
			if(fr->line[0] == '#') continue;
			const int nc = wx_split_tabs(fr->line, col, 24);
			if(nc < 17) continue;
			wx_hit_t h; memset(&h, 0, sizeof h);
			uint32_t id = hx_names_get(&W->names, col[0]);
			if(id == 0xFFFFFFFFu) continue;
			h.pb1 = id; h.dir1 = (uint8_t)(col[1][0] == '-');
			const int unprev1 = h.dir1 ? (int)W->pblen[id] - (int)(W->prev_off[id] + W->prev_len[id]) : (int)W->prev_off[id];          /* wtext.c:499-505 */
			h.tb = atoi(col[3]) + unprev1; h.te = atoi(col[4]) + unprev1;
			if((id = hx_names_get(&W->names, col[5])) == 0xFFFFFFFFu) continue;
			h.pb2 = id; h.dir2 = (uint8_t)(col[6][0] == '-');
			const int unprev2 = h.dir2 ? (int)W->pblen[id] - (int)(W->prev_off[id] + W->prev_len[id]) : (int)W->prev_off[id];
			h.qb = atoi(col[8]) + unprev2; h.qe = atoi(col[9]) + unprev2;
			h.cigar_in = strdup(col[16]);
			if(rn == rcap){ rcap = rcap ? rcap * 2 : 4096; rh = (wx_hit_t*)hx_realloc(rh, sizeof(wx_hit_t) * rcap); }
			rh[rn++] = h; W->n_in++;
		}
		if(rn >= o->block || (eof && rn)){
			W->t_read += wx_now() - t_rd;
			if(worker_on){ pthread_join(worker, NULL); worker_on = 0; }
			{ wx_hit_t *th = W->hits; const size_t tc = W->caphit; W->hits = rh; W->nhit = rn; W->caphit = rcap; rh = th; rcap = tc; rn = 0; }
			if(o->ncpu > 1 && !eof) worker_on = (pthread_create(&worker, NULL, wx_block_main, W) == 0);
			if(!worker_on) wx_process_block(W);
			t_rd = wx_now();
		}
	}
	if(worker_on) pthread_join(worker, NULL);
	free(rh);
	W->t_read += wx_now() - t_rd;
	hx_reader_close(fr);
	fprintf(stderr, "[%s] %llu overlaps read, %llu end extensions, %llu records written\n", wx_date(), W->n_in, W->n_ext, W->n_out);
	fprintf(stderr, "[wtext] host seconds (with -t > 1 the lines of a block are read beside the work on the block before): lines %.2f, clip + re-score %.2f (%d thread%s), extension sweeps %.2f, records %.2f\n", W->t_read, W->t_clip, o->ncpu > 1 ? (o->ncpu > 64 ? 64 : o->ncpu) : 1, o->ncpu > 1 ? "s" : "", W->t_ext, W->t_write);
	if(W->out != stdout) fclose(W->out); else fflush(stdout);
	free(col);
	return 0;
}

#endif
