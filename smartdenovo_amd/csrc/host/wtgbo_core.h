/*
 * wtgbo_core.h — the sequential side of `wtgbo` (wtgbo.c:120-208, 267-330, 366-611): options, inputs, the iteration over the
 * overlap graph, the order in which candidate pairs are aligned and their hits written.  The pair alignment itself
 * (align_hzmaux, hzm_aln.h:1684-1775) is NOT here: the includer defines gbo_align_jobs() — the device pipeline in wtgbo_main.c.
 *
 * Why the device can run ahead: inside one iteration the candidate lists depend only on the graph of the iteration and on the
 * closed-pair set, which the candidate walks themselves update (wtgbo.c:99-103, 254-258) — never on an alignment result.  So all
 * jobs of a pass are listed first, aligned as one stream of batches, and committed in list order.  The one place a result steers
 * the loop is wtgbo.c:190-191: in the graph pass a hit that covers the whole candidate ends the node's candidate list; the jobs
 * behind it were aligned for nothing and are dropped at commit (`-t 1` semantics, the order of the single worker).
 *
 * Quirks of the reference that are part of its output and are kept:
 *   - a hit that ends a node's list is written TWICE: the worker's `ret` flag survives the break, and the next wait_one (or the
 *     final close) prints the same hit again (wtgbo.c:185-187, 200) and pushes its edge again;
 *   - "contained" compares the hit's end on the NODE read with the length of the CANDIDATE read (wtgbo.c:51-54);
 *   - a new hit becomes an edge between the node read and ITSELF (wtgbo.c:130-131 sets both ends to id1), with the second
 *     length taken from the node read too;
 *   - -r is parsed and ignored; the chain threshold is -R (hzm_aln.h:1699); the window step is 0 (wtgbo.c:404,476).
 */
#ifndef WTGBO_CORE_H
#define WTGBO_CORE_H

#include <getopt.h>
#include <time.h>
#include "wtgbo_graph.h"
#include "wtz_ovlb.h"

/* every fatal input error leaves through _exit: the device context is being created on a helper thread meanwhile (wtgbo_main.c), and exit() would run the HIP
 * runtime's teardown under it (the same rule as DIE_NOW in wtzmo_main.c) */
#define GBO_DIE() do { fflush(NULL); _exit(1); } while(0)

typedef struct { char **a; int n, cap; } strlist_t;
static void sl_push(strlist_t *l, char *s){ if(l->n == l->cap){ l->cap = l->cap ? l->cap * 2 : 4; l->a = (char**)hx_realloc(l->a, sizeof(char*) * (size_t)l->cap); } l->a[l->n++] = s; }

typedef struct { uint32_t obj, qry, dir; uint32_t last_of_node; } gbo_job_t;        /* align read `qry` (strand `dir`) onto read `obj` */
typedef struct {
	int32_t ok;                    /* the device stages produced an alignment (windows, chain >= -R, at least one region) */
	int32_t score, tb, te, qb, qe, aln, mat, mis, ins, del;     /* kswx_t of global_align_regs_hzmo; tb/te on obj, qb/qe on the (oriented) qry */
	uint64_t cig_off; uint32_t cig_len;                        /* (len << 4 | op) words inside *cigar_pool */
} gbo_res_t;

typedef struct {
	int ncpu, w, W, ew, M, X, O, E, T, hz, zsize, kwin, kstep, zovl, zcut, kvar, refine;
	int min_score, margin, edgecov_cutoff, mat_score, max_ext, max_iter, overwrite;
	float min_id, best_score_cutoff;
	strlist_t pbs, ovls, obss, obts;
	char *output, *pairoutf;
	/* ours */
	int gpu; uint64_t pool_gb; uint32_t batch; int zindex_batch; int ingest_host; int binary_in;
} gbo_opt_t;

typedef struct {
	gbo_opt_t O;
	hx_store_t st; uint32_t n_rd; uint32_t *rdlen; hx_names_t names;
	gb_graph_t g; gb_closed_t closed; gb_biedges_t biedges;
	gbo_job_t *jobs; size_t njob, capjob;
	gbo_res_t *res; size_t capres;
	uint32_t *cigar_pool; uint64_t ncig, capcig;
	FILE *out;
	uint64_t n_aligned, n_dropped;
	void *backend;
} gbo_t;

/* defined by the includer: align jobs[0..n) and fill res[0..n) (cigars appended to G->cigar_pool through gbo_cigar_space) */
static void gbo_align_jobs(gbo_t *G, const gbo_job_t *jobs, size_t n, gbo_res_t *res);

static uint32_t *gbo_cigar_space(gbo_t *G, uint64_t n){
	if(G->ncig + n + 1 > G->capcig){ uint64_t c = G->capcig ? G->capcig : (1u << 20); while(c < G->ncig + n + 1) c += c / 2; G->cigar_pool = (uint32_t*)hx_realloc(G->cigar_pool, 4 * c); G->capcig = c; }
	uint32_t *p = G->cigar_pool + G->ncig; G->ncig += n; return p;
}

static const char *gbo_date(void){ static char buf[64]; time_t t = time(NULL); struct tm tmv; localtime_r(&t, &tmv); strftime(buf, sizeof buf, "%a %b %e %H:%M:%S %Y", &tmv); return buf; }

static int gbo_usage(void){      /* wtgbo.c:267-309 prints its usage to stdout and returns 1; the option LETTERS and defaults are the contract, the wording is ours */
	fputs(
	"WTGBO (MI355X tool set): overlapper on the overlap graph - aligns read pairs the graph proposes (align_hzmaux on gfx950)\n"
	"Usage: wtgbo [options]      (* = required, + = may be given several times)\n"
	" inputs    -i <reads fa/fq[.gz]> *+   -j <overlap file, >= 16 columns> *+   -b <name offset length: retained region> +   -L <name name: pairs already tested> +\n"
	" outputs   -o <new overlaps, - = stdout> *   -f overwrite   -9 <pairs tested, incl. those of -L>\n"
	" graph     -s <int> min score [200]   -m <float> min identity [0.6]   -u <int> max unaligned margin [100]   -c <int> min edge coverage [1]\n"
	"           -Q score = matches   -q <float> best-score cutoff [0.95]   -N <int> max iterations [5]\n"
	" seeding   -H no homopolymer compression   -z <int> z-mer size 5..16 [10]   -Z <int> max z-mer frequency [100]   -y <int> window [800]\n"
	"           -R <int> min seeded bases per window [200]   -r <int> (accepted, unused) [300]   -l <int> max z-mer length difference [2]\n"
	" alignment -M 2 -X -5 -O -3 -E -1 -T -50 scores   -w <int> band [50]   -e <int> extension band [800]   -W <int> max band [3200]   -n refine\n"
	" this build -t <int> accepted (the output is that of -t 1)   --gpu <id>   --pool-gb <n> [16]   --batch <pairs> [16384]   --zindex-batch <0|1>   --ingest <device|host>   --binary-in (-j: binary records of wtzmo --binary-out)\n",
	stdout);
	return 1;
}

static int gbo_parse_args(gbo_opt_t *o, int argc, char **argv){
	memset(o, 0, sizeof *o);
	o->min_score = 200; o->min_id = 0.6f; o->margin = 100; o->edgecov_cutoff = 1; o->mat_score = 0; o->best_score_cutoff = 0.95f;
	o->max_ext = 0; o->max_iter = 5; o->ncpu = 1; o->w = 50; o->ew = 800; o->W = 3200; o->M = 2; o->X = -5; o->O = -3; o->E = -1; o->T = -50;
	o->hz = 1; o->zsize = 10; o->kwin = 800; o->kstep = 0; o->zovl = 200; o->zcut = 100; o->kvar = 2; o->refine = 0;      /* wtgbo.c:385-411 */
	o->gpu = 0; o->pool_gb = 0; o->batch = 16384; o->zindex_batch = -1;
	static const struct option lopts[] = { {"gpu", 1, 0, 1001}, {"pool-gb", 1, 0, 1002}, {"batch", 1, 0, 1003}, {"zindex-batch", 1, 0, 1004}, {"ingest", 1, 0, 1005}, {"binary-in", 0, 0, 1006}, {0, 0, 0, 0} };
	int c; optind = 1;
	while((c = getopt_long(argc, argv, "hi:b:j:L:s:m:u:o:9:fQq:c:t:Hz:Z:y:l:r:R:w:e:W:M:X:O:E:T:nN:", lopts, NULL)) != -1){
		switch(c){
			case 'h': return 1;
			case 'i': sl_push(&o->pbs, optarg); break;
			case 'b': sl_push(&o->obts, optarg); break;
			case 'j': sl_push(&o->ovls, optarg); break;
			case 'L': sl_push(&o->obss, optarg); break;
			case 's': o->min_score = atoi(optarg); break;
			case 'm': o->min_id = (float)atof(optarg); break;
			case 'u': o->margin = atoi(optarg); break;
			case 'o': o->output = optarg; break;
			case '9': o->pairoutf = optarg; break;
			case 'f': o->overwrite = 1; break;
			case 'Q': o->mat_score = 1; break;
			case 'q': o->best_score_cutoff = (float)atof(optarg); break;
			case 'c': o->edgecov_cutoff = atoi(optarg); break;
			case 't': o->ncpu = atoi(optarg); break;
			case 'H': o->hz = 0; break;
			case 'z': o->zsize = atoi(optarg); break;
			case 'Z': o->zcut = atoi(optarg); break;
			case 'y': o->kwin = atoi(optarg); break;
			case 'l': o->kvar = atoi(optarg); break;
			case 'r': break;                                     /* wtgbo.c:443 sets ztot, which nothing reads */
			case 'R': o->zovl = (int)atof(optarg); break;
			case 'w': o->w = atoi(optarg); break;
			case 'e': o->ew = atoi(optarg); break;
			case 'W': o->W = atoi(optarg); break;
			case 'M': o->M = atoi(optarg); break;
			case 'X': o->X = atoi(optarg); break;
			case 'O': o->O = atoi(optarg); break;
			case 'E': o->E = atoi(optarg); break;
			case 'T': o->T = atoi(optarg); break;
			case 'n': o->refine = 1; break;
			case 'N': o->max_iter = atoi(optarg); break;
			case 1001: o->gpu = atoi(optarg); break;
			case 1002: o->pool_gb = (uint64_t)atoll(optarg); break;
			case 1003: o->batch = (uint32_t)atoi(optarg); if(o->batch < 1) o->batch = 1; break;
			case 1004: o->zindex_batch = atoi(optarg); break;
			case 1005: o->ingest_host = (strcmp(optarg, "host") == 0); break;      /* `device` (default): bases packed to 2 bits on the GPU; `host`: while reading */
			case 1006: o->binary_in = 1; break;                   /* the -j files are binary overlap streams (include/wtz_ovlb.h); regular files are recognised by their magic without it */
			default: return 1;
		}
	}
	if(o->output == NULL || o->pbs.n == 0 || o->ovls.n == 0) return 1;
	if(!o->overwrite && strcmp(o->output, "-")){ FILE *t = fopen(o->output, "r"); if(t){ fclose(t); fprintf(stderr, "File exists! '%s'\n\n", o->output); return 1; } }
	return 0;
}

/* split_string (string.h:251-274): tab-separated, empty fields vanish */
static int gbo_split_tabs(char *line, char **col, int maxcol){
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

static void gbo_load_inputs(gbo_t *G){
	gbo_opt_t *o = &G->O;
	hx_reader_t *fr = hx_reader_open(o->pbs.a, o->pbs.n);
	if(!fr) GBO_DIE();
	fprintf(stderr, "[%s] loading reads\n", gbo_date());
	hx_str_t name = {0, 0, 0}, seq = {0, 0, 0};
	while(hx_reader_seq(fr, &name, &seq)) hx_store_add(&G->st, name.s ? name.s : "", name.n, seq.s ? seq.s : "", seq.n);      /* file order = node id (wtgbo.c:459-467) */
	hx_reader_close(fr);
	G->n_rd = G->st.n_all;
	fprintf(stderr, "[%s] Done, %u reads\n", gbo_date(), G->n_rd);
	hx_names_build(&G->names, G->st.reads, G->n_rd);
	char *col[20];
	if(o->obts.n){             /* wtgbo.c:468-479, set_read_clip_strgraph wtlay.h:181-191 */
		fprintf(stderr, "[%s] loading the retained regions of the reads (-b)\n", gbo_date());
		if((fr = hx_reader_open(o->obts.a, o->obts.n)) == NULL) GBO_DIE();
		while(hx_reader_line(fr) != -1){
			const int nc = gbo_split_tabs(fr->line, col, 20);
			if(fr->line[0] == '#') continue;
			if(nc < 3) continue;
			const uint32_t id = hx_names_get(&G->names, col[0]);
			if(id == 0xFFFFFFFFu) continue;
			const int coff = atoi(col[1]), clen = atoi(col[2]);
			hx_read_t *rd = &G->st.reads[id];
			if(coff < 0 || coff + clen > (int)rd->len) continue;
			rd->off += (uint64_t)coff; rd->len = (uint32_t)clen;
		}
		hx_reader_close(fr);
		fprintf(stderr, "[%s] Done\n", gbo_date());
	} else fprintf(stderr, "[%s] no retained regions given\n", gbo_date());
	G->rdlen = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)G->n_rd + 1));
	for(uint32_t i = 0; i < G->n_rd; i++) G->rdlen[i] = G->st.reads[i].len;
	gb_graph_init(&G->g, G->n_rd, G->rdlen);
	G->g.min_score = o->min_score; G->g.min_id = o->min_id; G->g.max_margin = o->margin; G->g.mat_score = o->mat_score;
	gb_closed_init(&G->closed);
	if(o->obss.n){             /* wtgbo.c:488-506 */
		if((fr = hx_reader_open(o->obss.a, o->obss.n)) == NULL) GBO_DIE();
		fprintf(stderr, "[%s] loading the list of read pairs tested before (-L)\n", gbo_date());
		while(hx_reader_line(fr) != -1){
			const int nc = gbo_split_tabs(fr->line, col, 20);
			if(fr->line[0] == '#') continue;
			if(nc < 2) continue;
			const uint32_t a = hx_names_get(&G->names, col[0]); if(a == 0xFFFFFFFFu) continue;
			const uint32_t b = hx_names_get(&G->names, col[1]); if(b == 0xFFFFFFFFu) continue;
			gb_closed_put(&G->closed, gb_pair_id(a, b, 0));
		}
		hx_reader_close(fr);
		fprintf(stderr, "[%s] %llu pairs were tested before\n", gbo_date(), (unsigned long long)G->closed.slots.count);
	}
}

/* an accepted overlap enters the closed-pair set and, when it is a proper dovetail / containment, the edge list (wtlay.h:452-466) */
static void gbo_take_overlap(gbo_t *G, const gb_ovl_t *d, uint64_t *n){
	gb_biedge_t b;
	(*n)++;
	gb_closed_put(&G->closed, gb_pair_id(d->node[0], d->node[1], 0));
	gb_closed_put(&G->closed, gb_pair_id(d->node[0], d->node[1], 1));
	if(!gb_biedge_of(&G->g, d, &b, 1)) return;
	if(!gb_count_biedge(&G->g, &b)) return;
	gb_biedges_push(&G->biedges, &b);
}
/* f3: the same loader fed by binary records (include/wtz_ovlb.h; `bin/wtzmo --binary-out`) instead of text lines.  The ids of a record index the
 * WRITER's name table, so every name is looked up once per read, not twice per record; the identity the filter sees is the PRINTED one, i.e. the three
 * decimals `%0.3f` leaves of mat / aln, parsed back exactly like the text loader parses column 12.  Returns records read, -1 = not a binary stream. */
static long long gbo_load_overlaps_binary(gbo_t *G, FILE *fp, const char *first8, uint64_t *n){
	wtz_ovlb_reader_t rd;
	if(wtz_ovlb_open(&rd, fp, first8) != 0) return -1;
	uint32_t *map = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)rd.n_reads + 1));
	for(uint64_t i = 0; i < rd.n_reads; i++) map[i] = hx_names_get(&G->names, rd.names[i]);
	wtz_ovlb_rec_t r; gb_ovl_fields_t f; gb_ovl_t d; long long nrec = 0; int st; char idt[32];
	while((st = wtz_ovlb_next(&rd, &r)) == 1){
		nrec++;
		f.node[0] = map[r.id1]; f.node[1] = map[r.id2]; f.name[0] = rd.names[r.id1]; f.name[1] = rd.names[r.id2];
		f.dir[0] = 0; f.dir[1] = r.dir2 & 1; f.len[0] = (int)rd.rdlen[r.id1]; f.len[1] = (int)rd.rdlen[r.id2];
		f.beg[0] = r.tb; f.end[0] = r.te; f.beg[1] = r.qb; f.end[1] = r.qe; f.score = r.score; f.mat = r.mat;
		wtz_ovlb_identity_text(&r, idt); f.identity = (int)(atof(idt) * 1000);
		if(gb_accept_overlap(&G->g, &f, &d)) gbo_take_overlap(G, &d, n);
	}
	if(st < 0){ fprintf(stderr, " -- truncated or corrupt binary overlap stream after %lld records --\n", nrec); GBO_DIE(); }
	free(map); wtz_ovlb_close(&rd);
	return nrec;
}
/* the loader of the graph's overlaps (load_overlaps_strgraph, wtlay.h:443-468); every -j file may be text (>= 16 columns) or binary (sniffed by its magic) */
static void gbo_load_overlaps(gbo_t *G){
	char *col[20]; gb_ovl_fields_t f; gb_ovl_t d; uint64_t n = 0;
	for(int k = 0; k < G->O.ovls.n; k++){
		char *path = G->O.ovls.a[k];
		const size_t pl = strlen(path);
		const int is_stdin = (strcmp(path, "-") == 0), is_gz = (pl > 3 && strcmp(path + pl - 3, ".gz") == 0);
		if(G->O.binary_in || (!is_stdin && !is_gz)){
			FILE *fp = is_stdin ? stdin : fopen(path, "rb");
			if(!fp){ fprintf(stderr, " -- Cannot open %s --\n", path); GBO_DIE(); }
			char m8[8]; const size_t got = fread(m8, 1, 8, fp);
			if(got == 8 && memcmp(m8, WTZ_OVLB_MAGIC, 8) == 0){
				const long long nrec = gbo_load_overlaps_binary(G, fp, m8, &n);
				if(nrec < 0){ fprintf(stderr, " -- %s: unreadable binary overlap header --\n", path); GBO_DIE(); }
				if(!is_stdin) fclose(fp);
				continue;
			}
			if(G->O.binary_in){ fprintf(stderr, " -- %s is not a binary overlap stream (--binary-in) --\n", path); GBO_DIE(); }
			fclose(fp);       /* text: read it again through the line reader */
		}
		hx_reader_t *fr = hx_reader_open(&path, 1);
		if(!fr) GBO_DIE();
		while(hx_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			const int nc = gbo_split_tabs(fr->line, col, 20);
			if(!gb_fields_from_text(&G->names, col, nc, &f)) continue;
			if(gb_accept_overlap(&G->g, &f, &d)) gbo_take_overlap(G, &d, &n);
		}
		hx_reader_close(fr);
	}
	fprintf(stderr, "loaded %llu overlaps\n", (unsigned long long)n);
	gb_build_edges(&G->g, &G->biedges);
}

static void gbo_job_push(gbo_t *G, uint32_t obj, uint32_t qry, uint32_t dir){
	if(G->njob == G->capjob){ G->capjob = G->capjob ? G->capjob * 2 : 4096; G->jobs = (gbo_job_t*)hx_realloc(G->jobs, sizeof(gbo_job_t) * G->capjob); }
	gbo_job_t j; j.obj = obj; j.qry = qry; j.dir = dir; j.last_of_node = 0; G->jobs[G->njob++] = j;
}

/* the gates of align_hzmaux behind the stitched alignment (hzm_aln.h:1715-1718): the hit must cover min_sm of the overlap its
 * ends imply.  tlen = length of obj, qlen = length of qry */
static int gbo_hit_passes(const gbo_res_t *r, int tlen, int qlen, float min_sm, int refined){
	if(!r->ok) return 0;
	if(refined) return 1;       /* -n: the gates apply to the alignment BEFORE kswx_refine_alignment (hzm_aln.h:1718 vs 1721-1729) and ran on the device */
	int beg = r->qb - r->tb; if(beg < 0) beg = 0;
	int end = r->qe + tlen - r->te; if(end > qlen) end = qlen;
	const int ovl = end - beg;
	if(r->score < 0 || (float)r->mat < (float)r->aln * min_sm || (float)r->mat < (float)ovl * min_sm) return 0;
	return 1;
}

static inline size_t gbo_put_int(char *o, long long v){ char t[24]; int n = 0; unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v; size_t k = 0; if(v < 0) o[k++] = '-'; do { t[n++] = (char)('0' + u % 10); u /= 10; } while(u); while(n) o[k++] = t[--n]; return k; }

/* output_wtgbo, wtgbo.c:120-141: the 17 columns + the hit as an edge of the next iteration's graph */
static int gbo_output(gbo_t *G, const gbo_job_t *j, const gbo_res_t *r){
	const hx_read_t *R = G->st.reads;
	fprintf(G->out, "%s\t%c\t%d\t%d\t%d", R[j->obj].name, '+', (int)G->rdlen[j->obj], r->tb, r->te);
	fprintf(G->out, "\t%s\t%c\t%d\t%d\t%d", R[j->qry].name, "+-"[j->dir], (int)G->rdlen[j->qry], r->qb, r->qe);
	fprintf(G->out, "\t%d\t%0.3f\t%d\t%d\t%d\t%d\t", r->score, 1.0 * r->mat / r->aln, r->mat, r->mis, r->ins, r->del);
	{       /* kswx_print_cigars, kswx.h:54-63: every word, zero lengths included */
		static char *buf = NULL; static size_t cap = 0;
		const size_t need = (size_t)r->cig_len * 12 + 2;
		if(need > cap){ cap = need * 2; buf = (char*)hx_realloc(buf, cap); }
		size_t k = 0; const uint32_t *cg = G->cigar_pool + r->cig_off;
		for(uint32_t i = 0; i < r->cig_len; i++){ k += gbo_put_int(buf + k, (long long)(cg[i] >> 4)); buf[k++] = "MIDX????????????"[cg[i] & 0xF]; }
		buf[k++] = '\n';
		fwrite(buf, 1, k, G->out);
	}
	gb_ovl_t O; gb_biedge_t B;
	O.node[0] = j->obj; O.node[1] = j->obj;                       /* wtgbo.c:130-131 */
	O.dir[0] = 0; O.dir[1] = (int)j->dir;
	O.beg[0] = r->tb; O.beg[1] = r->qb; O.end[0] = r->te; O.end[1] = r->qe;
	O.score = r->score; O.identity = (int)(1000.0 * r->mat / r->aln);
	if(gb_biedge_of(&G->g, &O, &B, 0)){ gb_biedges_push(&G->biedges, &B); return 1; }
	return 0;
}

/* align the listed jobs in batches and commit them in order.  graph_pass: wtgbo.c:184-197 (break on a containing hit, the
 * repeated line); otherwise wtgbo.c:310-320.  Returns the number of hits that became edges. */
static uint64_t gbo_run_jobs(gbo_t *G, int graph_pass){
	const gbo_opt_t *o = &G->O;
	uint64_t ret = 0;
	if(G->njob > G->capres){ G->capres = G->njob; G->res = (gbo_res_t*)hx_realloc(G->res, sizeof(gbo_res_t) * G->capres); }
	/* worker state of the reference's single thread */
	int w_ret = 0, w_contained = 0; size_t w_job = 0;
	size_t done = 0;            /* results available for jobs [0, done) */
	size_t i = 0;
	G->ncig = 0;
	while(i < G->njob){
		while(i >= done){
			if(i > done) done = i;       /* jobs dropped behind a containing hit are not aligned at all when they fall into a later batch */
			size_t n = G->njob - done; if(n > o->batch) n = o->batch;
			G->ncig = 0;         /* the CIGARs of committed batches are no longer needed (the pending hit is re-pointed below) */
			gbo_res_t keep; uint32_t *keepc = NULL; memset(&keep, 0, sizeof keep);
			if(w_ret){ keep = G->res[w_job]; keepc = (uint32_t*)hx_realloc(NULL, 4 * ((size_t)keep.cig_len + 1)); memcpy(keepc, G->cigar_pool + keep.cig_off, 4 * (size_t)keep.cig_len); }
			gbo_align_jobs(G, G->jobs + done, n, G->res + done);
			if(w_ret){ uint32_t *p = gbo_cigar_space(G, keep.cig_len); memcpy(p, keepc, 4 * (size_t)keep.cig_len); G->res[w_job].cig_off = (uint64_t)(p - G->cigar_pool); free(keepc); }
			G->n_aligned += n;
			done += n;
		}
		const gbo_job_t *j = &G->jobs[i];
		/* thread_wait_one; if(mgbo->ret) output (wtgbo.c:185-187 / 311-312) */
		if(w_ret) ret += (uint64_t)gbo_output(G, &G->jobs[w_job], &G->res[w_job]);
		if(graph_pass){
			const int first_of_node = (i == 0) || G->jobs[i - 1].last_of_node;
			if(!first_of_node && w_contained){
				/* break (wtgbo.c:190-191): the rest of this node's candidates is never aligned; `ret` stays set */
				while(!G->jobs[i].last_of_node){ i++; G->n_dropped++; }
				i++; G->n_dropped++;
				continue;
			}
		}
		/* the worker runs job i (wtgbo.c:37-56) */
		const gbo_res_t *r = &G->res[i];
		w_job = i;
		w_ret = gbo_hit_passes(r, (int)G->rdlen[j->obj], (int)G->rdlen[j->qry], o->min_id, o->refine);
		w_contained = (w_ret && r->tb == 0 && r->te == (int)G->rdlen[j->qry]);
		i++;
	}
	/* thread_beg_close: if(mgbo->ret) output (wtgbo.c:200 / 326) */
	if(w_ret) ret += (uint64_t)gbo_output(G, &G->jobs[w_job], &G->res[w_job]);
	return ret;
}

static int gbo_run(gbo_t *G){
	gbo_opt_t *o = &G->O;
	gb_graph_t *g = &G->g;
	G->out = strcmp(o->output, "-") ? fopen(o->output, "w") : stdout;
	if(!G->out){ fprintf(stderr, "Cannot open %s for write\n", o->output); return 1; }
	gb_heap_t heap; memset(&heap, 0, sizeof heap);
	gb_u32v_t c32; memset(&c32, 0, sizeof c32);
	gb_u64v_t c64; memset(&c64, 0, sizeof c64);
	gb_marks_t marks; memset(&marks, 0, sizeof marks);
	gb_uu_t tab; gb_uu_init(&tab, 1023);            /* init_uuhash(1023), wtgbo.c:283 */
	int iter = 0, oob = 0;
	while(iter < o->max_iter){
		iter++;
		fprintf(stderr, "---------------------------\n[%s] iteration %d\n", gbo_date(), iter);
		memset(g->dead, 0, (size_t)g->n_rd + 1);
		if(iter == 1){
			fprintf(stderr, "[%s] loading the overlaps (-j)\n", gbo_date());
			gbo_load_overlaps(G);
		} else {              /* wtgbo.c:540-553 */
			fprintf(stderr, "[%s] rebuilding the edge lists with the new overlaps\n", gbo_date());
			for(uint32_t i = 0; i < g->n_rd; i++){ g->nodes[i].ecnt[0] = g->nodes[i].ecnt[1] = 0; }
			gb_biedges_t kept; memset(&kept, 0, sizeof kept);
			for(size_t i = 0; i < G->biedges.n; i++) if(gb_count_biedge(g, &G->biedges.a[i])) gb_biedges_push(&kept, &G->biedges.a[i]);
			if(kept.n != G->biedges.n) fprintf(stderr, "[wtgbo-mi355x] %llu overlaps beyond %u edges of a read side are left out (the reference writes them past the side's slice)\n", (unsigned long long)(G->biedges.n - kept.n), GB_MAX_EDGE);
			gb_build_edges(g, &kept);
			free(kept.a);
		}
		fprintf(stderr, "[%s] edge coverage\n", gbo_date());
		gb_edge_coverage(g);
		unsigned long long n = gb_drop_duplicate_edges(g);
		fprintf(stderr, "[%s] %llu duplicate edges dropped\n", gbo_date(), n);
		n = gb_mask_contained(g);
		fprintf(stderr, "[%s] %llu contained reads masked\n", gbo_date(), n);
		n = gb_mask_low_cov(g, (uint32_t)o->edgecov_cutoff);
		fprintf(stderr, "[%s] %llu edges below coverage %u masked\n", gbo_date(), n, (unsigned)o->edgecov_cutoff);
		n = gb_best_overlap(g, o->best_score_cutoff);
		fprintf(stderr, "[%s] best-overlap rule: %llu other edges cut\n", gbo_date(), n);
		/* ---- graph based overlapping (gbo_core_wtgbo, wtgbo.c:143-208) ---- */
		fprintf(stderr, "[%s] pass 1: pairs two steps apart in the graph\n", gbo_date());
		G->njob = 0;
		for(uint32_t node = 0; node < g->n_rd; node++){
			if(g->dead[node]) continue;
			if(g->nodes[node].mutual[0] && g->nodes[node].mutual[1]) continue;
			c32.n = 0;
			gb_graph_candidates(g, node, (uint32_t)o->max_ext, &G->closed, &c32, &heap, &oob);
			if(c32.n == 0) continue;
			for(size_t i = 0; i < c32.n; i++) gbo_job_push(G, node, c32.a[i] >> 1, c32.a[i] & 1u);
			G->jobs[G->njob - 1].last_of_node = 1;
		}
		fprintf(stderr, "[%s] %llu candidates\n", gbo_date(), (unsigned long long)G->njob);
		unsigned long long nn = gbo_run_jobs(G, 1);
		fprintf(stderr, "[%s] %llu new overlaps\n", gbo_date(), nn);
		/* ---- anchoring based overlapping (abo_core_wtgbo, wtgbo.c:267-330) ---- */
		fprintf(stderr, "[%s] pass 2: pairs anchored on the same stretch of a read\n", gbo_date());
		G->njob = 0;
		for(uint32_t node = 0; node < g->n_rd; node++){
			c64.n = 0;
			gb_anchor_candidates(g, node, &G->closed, &c64, &marks, &tab);
			for(size_t i = 0; i < c64.n; i++) gbo_job_push(G, (uint32_t)(c64.a[i] >> 33), (uint32_t)((c64.a[i] >> 1) & 0xFFFFFFFFu), (uint32_t)(c64.a[i] & 1u));
		}
		fprintf(stderr, "[%s] %llu candidates\n", gbo_date(), (unsigned long long)G->njob);
		n = gbo_run_jobs(G, 0);
		fprintf(stderr, "[%s] %llu new overlaps\n", gbo_date(), n);
		nn += n;
		fflush(G->out);
		if(nn == 0) break;
	}
	if(oob) fprintf(stderr, "[wtgbo-mi355x] note: the graph walk read one element past the edge array (wtgbo.c:91), as the reference does; a zero edge was used\n");
	if(G->out != stdout) fclose(G->out);
	if(o->pairoutf){            /* wtgbo.c:590-601: slot order of the closed-pair table */
		FILE *po = fopen(o->pairoutf, "w");
		if(po){
			const hx_refslots_t *t = &G->closed.slots;
			for(uint64_t s = 0; s < t->size; s++){
				if(!t->full[s]) continue;
				const uint32_t id1 = (uint32_t)(t->slot[s] >> 33), id2 = (uint32_t)((t->slot[s] & 0xFFFFFFFFu) >> 1);
				fprintf(po, "%s\t%s\n", G->st.reads[id1].name, G->st.reads[id2].name);
			}
			fclose(po);
		}
	}
	fprintf(stderr, "[wtgbo-mi355x] %llu pairs aligned on the device, %llu of them dropped behind a containing hit\n", (unsigned long long)G->n_aligned, (unsigned long long)G->n_dropped);
	return 0;
}

#endif
