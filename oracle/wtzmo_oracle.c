/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ora_util.h).
 *
 * wtzmo_oracle — single-threaded CPU restatement of `wtzmo -t 1` with the reference's
 * command line (reference wtzmo.c:1512-1812: getopt string 1594, defaults 1543-1588,
 * validation 1653-1661, loaders 1691-1773, side files 1781-1804).  `-t` is accepted and
 * ignored: the oracle always runs the deterministic one-worker order.
 *
 * Extra (oracle-only) option:  --stats <file>   writes "pairs\tpair_bp\tseconds" of the
 * overlap phase (SURVEY §8d metric numerator) for bench.py's cpu_baseline leg.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <unistd.h>
#include <time.h>
#include "ora_overlap.h"

static int usage(void){
	printf("wtzmo_oracle: CPU restatement of SMARTdenovo wtzmo (-t 1 semantics); same options as wtzmo\n"
	       "Usage: wtzmo_oracle -i <reads> -o <out|-> [wtzmo options]\n");
	return 1;
}

typedef struct { char **a; int n, cap; } strlist_t;
static void sl_push(strlist_t *l, char *s){ if(l->n == l->cap){ l->cap = l->cap ? l->cap * 2 : 4; l->a = (char**)ora_xrealloc(l->a, sizeof(char*) * (size_t)l->cap); } l->a[l->n++] = s; }

/* name -> id map (cuhash in the reference; last put wins on duplicate names like kv_put) */
typedef struct { uint32_t *tab; size_t cap; const ora_read_t *reads; } namemap_t;
static uint64_t name_hash(const char *s){ uint64_t h = 1469598103934665603ULL; while(*s){ h ^= (unsigned char)*s++; h *= 1099511628211ULL; } return h; }
static void namemap_build(namemap_t *m, const ora_read_t *reads, uint32_t n){
	m->cap = 16; while(m->cap < (size_t)n * 2 + 2) m->cap <<= 1;
	m->tab = (uint32_t*)ora_xrealloc(NULL, m->cap * 4); memset(m->tab, 0xFF, m->cap * 4); m->reads = reads;
	for(uint32_t i = 0; i < n; i++){
		size_t k = name_hash(reads[i].name) & (m->cap - 1);
		while(m->tab[k] != 0xFFFFFFFFu && strcmp(reads[m->tab[k]].name, reads[i].name)) k = (k + 1) & (m->cap - 1);
		m->tab[k] = i;
	}
}
static uint32_t namemap_get(const namemap_t *m, const char *s){
	size_t k = name_hash(s) & (m->cap - 1);
	while(m->tab[k] != 0xFFFFFFFFu){ if(strcmp(m->reads[m->tab[k]].name, s) == 0) return m->tab[k]; k = (k + 1) & (m->cap - 1); }
	return 0xFFFFFFFFu;
}

static int cmp_u64(const void *a, const void *b){ uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : (x > y); }

static int split_tabs(char *line, char **cols, int maxc){
	int n = 0; char *p = line;
	while(n < maxc){ cols[n++] = p; while(*p && *p != '\t' && *p != ' ') p++; if(!*p) break; *p++ = 0; }
	return n;
}

int main(int argc, char **argv){
	ora_ctx_t *C = (ora_ctx_t*)calloc(1, sizeof(ora_ctx_t));
	ora_params_t *P = &C->P;
	strlist_t pbs = {0}, flts = {0}, ovls = {0}, obts = {0}, tbas = {0};
	char *output = NULL, *pairoutf = NULL, *statsf = NULL;
	int c, min_rdlen = 0, overwrite = 0, n_job = 1, i_job = 0, dot_matrix = 0, skip_contained_flag = 1;
	float optval;
	ora_params_default(P);
	static struct option lopts[] = { {"stats", required_argument, 0, 1000}, {0, 0, 0, 0} };
	while((c = getopt_long(argc, argv, "ht:P:p:Ni:b:J:I:o:9:S:fCH:k:G:z:Z:U:y:d:r:q:l:K:A:B:r:R:L:F:W:w:e:M:X:O:E:T:s:m:nv", lopts, NULL)) != -1){
		switch(c){
			case 1000: statsf = optarg; break;
			case 'h': return usage();
			case 't': break;
			case 'P': n_job = atoi(optarg); break;
			case 'p': i_job = atoi(optarg); break;
			case 'N': P->do_align = 0; break;
			case 'i': sl_push(&pbs, optarg); break;
			case 'b': sl_push(&obts, optarg); break;
			case 'J': min_rdlen = atoi(optarg); break;
			case 'I': sl_push(&tbas, optarg); break;
			case 'o': output = optarg; break;
			case '9': pairoutf = optarg; break;
			case 'S': P->ksave = (uint32_t)atoi(optarg); if(atoi(optarg) < 1) return usage(); break;
			case 'f': overwrite = 1; break;
			case 'C': skip_contained_flag = 0; break;
			case 'H': { int hk = atoi(optarg); P->hz = (hk >> 1) & 1; P->hk = hk & 1; } break;
			case 'k': P->ksize = (uint32_t)atoi(optarg); break;
			case 'K': P->max_kmer_freq = (uint32_t)atoi(optarg); break;
			case 'z': P->zsize = (uint32_t)atoi(optarg); break;
			case 'Z': P->max_zmer_freq = (uint32_t)atoi(optarg); break;
			case 'U': optval = (float)atof(optarg);
				if(optval < 0){ dot_matrix = 5; break; }
				switch(dot_matrix){
					case 0: P->xvar = (int)optval; break;
					case 1: P->yvar = (int)optval; break;
					case 2: P->min_block_len = (int)optval; break;
					case 3: P->deviation_penalty = optval; break;
					case 4: P->gap_penalty = optval; break;
					default: dot_matrix = 5;
				}
				dot_matrix++;
				break;
			case 'y': P->kwin = (uint32_t)atoi(optarg); break;
			case 'l': P->max_kmer_var = (uint32_t)atoi(optarg); break;
			case 'd': P->kovl = (uint32_t)(int)atof(optarg); break;
			case 'G': P->n_idx = (uint32_t)atoi(optarg); break;
			case 'r': P->ztot = (uint32_t)(int)atof(optarg); break;
			case 'R': P->zovl = (uint32_t)(int)atof(optarg); break;
			case 'q': P->win_rep_cutoff = (float)atoi(optarg); break;
			case 'A': P->ncand = (uint32_t)atoi(optarg); break;
			case 'B': P->nbest = (uint32_t)atoi(optarg); break;
			case 'w': P->w = atoi(optarg); break;
			case 'e': P->ew = atoi(optarg); break;
			case 'W': P->W = atoi(optarg); break;
			case 'M': P->M = atoi(optarg); break;
			case 'X': P->X = atoi(optarg); break;
			case 'O': P->O = atoi(optarg); break;
			case 'E': P->E = atoi(optarg); break;
			case 'T': P->T = atoi(optarg); break;
			case 'L': sl_push(&ovls, optarg); break;
			case 'F': sl_push(&flts, optarg); break;
			case 's': P->min_score = atoi(optarg); break;
			case 'm': P->min_id = (float)atof(optarg); break;
			case 'n': P->refine = 1; break;
			case 'v': break;
			default: return usage();
		}
	}
	if(output == NULL) return usage();
	if(!overwrite && strcmp(output, "-") && access(output, F_OK) == 0){ fprintf(stderr, "File exists! '%s'\n\n", output); return usage(); }
	if(pbs.n == 0) return usage();
	if(P->ksize > 32 || P->ksize < 5) return usage();
	if(P->zsize > 16 || P->zsize < 5) return usage();
	P->max_overhang = 2 * P->xvar;
	P->kstep = P->kwin / 2;
	P->dot_matrix = dot_matrix;

	ora_reader_t *fr = ora_reader_open(pbs.a, pbs.n);
	if(fr == NULL){ fprintf(stderr, " -- Cannot open %s --\n", pbs.a[0]); exit(1); }
	vec_u8 name = {0}, seq = {0};
	while(ora_reader_seq(fr, &name, &seq)){
		if((int)seq.n < min_rdlen) continue;
		ora_store_add_read(&C->st, (char*)name.a, name.n, (char*)seq.a, seq.n);
	}
	ora_reader_close(fr);
	ora_sort_reads_by_len(C->st.reads.a, C->st.reads.n, NULL);
	if(tbas.n){
		if((fr = ora_reader_open(tbas.a, tbas.n)) == NULL) exit(1);
		while(ora_reader_seq(fr, &name, &seq)){
			if((int)seq.n < min_rdlen) continue;
			ora_store_add_read(&C->st, (char*)name.a, name.n, (char*)seq.a, seq.n);
			C->st.n_rd--; C->n_qr++;
		}
		ora_reader_close(fr);
	}
	uint32_t n_all = C->st.n_rd + C->n_qr;
	C->masked = (uint8_t*)calloc((size_t)n_all + 1, 1);
	C->rdcovs = (uint32_t*)calloc((size_t)n_all + 1, 4);
	namemap_t nm; namemap_build(&nm, C->st.reads.a, C->st.n_rd);   /* only -i reads are named (wtzmo.c:1709-1711) */
	char *cols[8];
	if(obts.n){
		if((fr = ora_reader_open(obts.a, obts.n)) == NULL) exit(1);
		while(ora_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			if(split_tabs(fr->line, cols, 8) < 3) continue;
			uint32_t id = namemap_get(&nm, cols[0]); int coff = atoi(cols[1]), clen = atoi(cols[2]);
			if(id == 0xFFFFFFFFu) continue;
			ora_read_t *rd = &C->st.reads.a[id];
			if(coff < 0 || coff + clen > (int)rd->len) continue;
			rd->off += (uint64_t)coff; rd->len = (uint32_t)clen;
		}
		ora_reader_close(fr);
	}
	if(flts.n){
		if((fr = ora_reader_open(flts.a, flts.n)) == NULL) exit(1);
		while(ora_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			uint32_t id = namemap_get(&nm, fr->line);
			if(id == 0xFFFFFFFFu) continue;
			C->masked[id] = 1;
		}
		ora_reader_close(fr);
	}
	if(ovls.n){
		if((fr = ora_reader_open(ovls.a, ovls.n)) == NULL) exit(1);
		while(ora_reader_line(fr) != -1){
			if(fr->line[0] == '#') continue;
			if(split_tabs(fr->line, cols, 8) < 2) continue;
			uint32_t a = namemap_get(&nm, cols[0]), b = namemap_get(&nm, cols[1]);
			if(a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) continue;
			ora_u64set_put(&C->closed, ora_pair_key(a, b));
		}
		ora_reader_close(fr);
	}
	FILE *out = strcmp(output, "-") ? fopen(output, "w") : stdout;
	struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
	uint64_t nrec = ora_overlap_all(C, (uint32_t)n_job, (uint32_t)i_job, out);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if(strcmp(output, "-")) fclose(out);
	double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	fprintf(stderr, "wtzmo_oracle: %u reads, %llu records, %llu pairs, %llu pair-bp, %.3f s\n", C->st.n_rd, (unsigned long long)nrec,
		(unsigned long long)C->n_pairs, (unsigned long long)C->pair_bp, secs);
	if(statsf){ FILE *sf = fopen(statsf, "w"); if(sf){ fprintf(sf, "%llu\t%llu\t%.6f\n", (unsigned long long)C->n_pairs, (unsigned long long)C->pair_bp, secs); fclose(sf); } }
	if(skip_contained_flag && strcmp(output, "-")){
		char *maskf = (char*)ora_xrealloc(NULL, strlen(output) + 16);
		sprintf(maskf, "%s.contained", output);
		FILE *mf = fopen(maskf, "w");
		for(uint32_t i = 0; i < C->st.n_rd; i++){ if(C->masked[i]) fprintf(mf, "%s\n", C->st.reads.a[i].name); }
		fclose(mf); free(maskf);
	}
	if(pairoutf){
		/* the reference writes the pairs in its hash-table iteration order; the oracle writes them sorted (set equality is the contract) */
		FILE *pf = fopen(pairoutf, "w");
		vec_u64 all = {0};
		for(size_t i = 0; i < C->closed.cap; i++) if(C->closed.tab[i] != ~0ULL) vec_u64_push(&all, C->closed.tab[i]);
		qsort(all.a, all.n, 8, cmp_u64);
		for(size_t i = 0; i < all.n; i++){
			uint32_t a = (uint32_t)(all.a[i] >> 33), b = (uint32_t)((all.a[i] & 0xFFFFFFFFu) >> 1);
			fprintf(pf, "%s\t%s\n", C->st.reads.a[a].name, C->st.reads.a[b].name);
		}
		fclose(pf);
	}
	return 0;
}
