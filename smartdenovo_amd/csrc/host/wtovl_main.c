/*
 * wtovl — binary overlap records (include/wtz_ovlb.h; `wtzmo --binary-out`) back to the text the reference's consumers parse
 * (SURVEY 8f3): a consumer that stays on text - `wtclp` (load_alignments_wtclp, wtclp.c:111-180: columns 0-4, 5-9, 11 of >= 12) or
 * `wtlay` (parse_overlap_item_strgraph, wtlay.h:238-268: columns 0-12 of >= 16) - reads `wtovl x.ovlb |` instead of the file.
 *
 *   wtovl [-c 16|17] [in.ovlb | -]      16 columns: byte-identical to `cut -f1-16` of the text output (what the zmo pipeline keeps,
 *                                        smartdenovo.pl:58); 17: the same + a CIGAR column "0M" (the dmo engine's form, wtzmo.c:1243)
 *   wtovl -s [in.ovlb | -]               summary only: reads in the name table, records
 *   wtovl -b [in.ovl | -] > out.ovlb     the other way: an existing text file (16 or 17 columns, '#' lines skipped) as a binary stream, for consumers that
 *                                        read binary (wtgbo --binary-in; wtlay / wtclp with the loader patch of integration/).  The name table is the reads in
 *                                        order of first appearance; `aln` is mat+mis+ins+del where that prints the file's identity column, else the
 *                                        nearest denominator that does (dmo records), else the line is refused
 * Host text filter, no arithmetic in it, no device.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wtz_ovlb.h"

static int usage(void){
	fputs("WTOVL: binary overlap records (wtzmo --binary-out) -> the reference's text columns\n"
	      "Usage: wtovl [-c 16|17] [-s] [file | -]\n"
	      "       wtovl -b [text file | -] > binary      (text records -> binary stream)\n", stdout);
	return 1;
}

/* ---- text -> binary ---- */
typedef struct { char *name; uint32_t len, id; } tb_read_t;
static tb_read_t *tb_tab = NULL; static size_t tb_cap = 0, tb_n = 0; static uint32_t *tb_order = NULL;      /* open addressing by name; tb_order: slot of read id */
static uint64_t tb_hash(const char *s){ uint64_t h = 1469598103934665603ULL; while(*s){ h ^= (unsigned char)*s++; h *= 1099511628211ULL; } return h; }
static void tb_grow(void){
	const size_t nc = tb_cap ? tb_cap * 2 : 1024; tb_read_t *nt = (tb_read_t*)calloc(nc, sizeof(tb_read_t));
	if(!nt){ fprintf(stderr, " -- out of memory --\n"); exit(1); }
	for(size_t i = 0; i < tb_cap; i++) if(tb_tab[i].name){ size_t k = tb_hash(tb_tab[i].name) & (nc - 1); while(nt[k].name) k = (k + 1) & (nc - 1); nt[k] = tb_tab[i]; }
	free(tb_tab); tb_tab = nt; tb_cap = nc;
	tb_order = (uint32_t*)realloc(tb_order, sizeof(uint32_t) * nc);
	for(size_t k = 0; k < nc; k++) if(nt[k].name) tb_order[nt[k].id] = (uint32_t)k;
}
static uint32_t tb_id(const char *name, uint32_t len){
	if((tb_n + 1) * 2 > tb_cap) tb_grow();
	size_t k = tb_hash(name) & (tb_cap - 1);
	while(tb_tab[k].name){
		if(!strcmp(tb_tab[k].name, name)){ if(tb_tab[k].len != len){ fprintf(stderr, " -- read %s has two lengths (%u, %u) --\n", name, tb_tab[k].len, len); exit(1); } return tb_tab[k].id; }
		k = (k + 1) & (tb_cap - 1);
	}
	tb_tab[k].name = strdup(name); tb_tab[k].len = len; tb_tab[k].id = (uint32_t)tb_n; tb_order[tb_n] = (uint32_t)k;
	return (uint32_t)tb_n++;
}
static int text_to_binary(FILE *fp){
	char *line = NULL; size_t cap = 0; ssize_t n;
	wtz_ovlb_rec_t *recs = NULL; size_t nrec = 0, crec = 0; unsigned long long lineno = 0;
	while((n = getline(&line, &cap, fp)) >= 0){
		lineno++;
		if(n && line[n - 1] == '\n') line[--n] = 0;
		if(line[0] == '#' || line[0] == 0) continue;
		char *col[17]; int nc = 0; char *p = line;
		while(nc < 17){ col[nc++] = p; char *t = strchr(p, '\t'); if(!t) break; *t = 0; p = t + 1; }
		if(nc < 16){ fprintf(stderr, " -- line %llu has %d columns (16 needed) --\n", lineno, nc); return 1; }
		wtz_ovlb_rec_t r; memset(&r, 0, sizeof r);
		r.id1 = tb_id(col[0], (uint32_t)atoi(col[2])); r.id2 = tb_id(col[5], (uint32_t)atoi(col[7]));
		if(col[1][0] != '+'){ fprintf(stderr, " -- line %llu: read 1 is not on '+' --\n", lineno); return 1; }
		r.dir2 = col[6][0] == '-'; r.tb = atoi(col[3]); r.te = atoi(col[4]); r.qb = atoi(col[8]); r.qe = atoi(col[9]); r.score = atoi(col[10]);
		r.mat = atoi(col[12]); r.mis = atoi(col[13]); r.ins = atoi(col[14]); r.del = atoi(col[15]);
		/* the denominator behind the identity column: the printed three decimals must come out again */
		char idt[32]; long long sum = (long long)r.mat + r.mis + r.ins + r.del; int ok = 0;
		r.aln = (uint32_t)(sum > 0 ? sum : 1); wtz_ovlb_identity_text(&r, idt); ok = !strcmp(idt, col[11]);
		if(!ok){
			const double idv = atof(col[11]); const long long guess = idv > 0 ? (long long)(r.mat / idv + 0.5) : 1;
			for(long long d = 0; d < 4096 && !ok; d++) for(int sg = -1; sg <= 1 && !ok; sg += 2){
				const long long a = guess + sg * d; if(a < 1) continue;
				r.aln = (uint32_t)a; wtz_ovlb_identity_text(&r, idt); ok = !strcmp(idt, col[11]);
			}
		}
		if(!ok){ fprintf(stderr, " -- line %llu: no alignment length prints the identity %s for %d matches --\n", lineno, col[11], r.mat); return 1; }
		if(nrec == crec){ crec = crec ? crec * 2 : 4096; recs = (wtz_ovlb_rec_t*)realloc(recs, crec * sizeof *recs); if(!recs){ fprintf(stderr, " -- out of memory --\n"); return 1; } }
		recs[nrec++] = r;
	}
	free(line);
	const char **names = (const char**)malloc(sizeof(char*) * (tb_n + 1)); uint32_t *lens = (uint32_t*)malloc(4 * (tb_n + 1));
	for(size_t i = 0; i < tb_n; i++){ names[i] = tb_tab[tb_order[i]].name; lens[i] = tb_tab[tb_order[i]].len; }
	if(wtz_ovlb_write_header(stdout, tb_n, names, lens) != 0 || (nrec && fwrite(recs, sizeof *recs, nrec, stdout) != nrec) || fflush(stdout) != 0){ fprintf(stderr, " -- write error --\n"); return 1; }
	return 0;
}

int main(int argc, char **argv){
	int ncol = 16, summary = 0, to_binary = 0; const char *path = "-";
	for(int i = 1; i < argc; i++){
		if(!strcmp(argv[i], "-c") && i + 1 < argc){ ncol = atoi(argv[++i]); if(ncol != 16 && ncol != 17) return usage(); }
		else if(!strcmp(argv[i], "-s")) summary = 1;
		else if(!strcmp(argv[i], "-b")) to_binary = 1;
		else if(!strcmp(argv[i], "-h")) return usage();
		else path = argv[i];
	}
	FILE *fp = strcmp(path, "-") ? fopen(path, "rb") : stdin;
	if(!fp){ fprintf(stderr, " -- Cannot open %s --\n", path); return 1; }
	if(to_binary) return text_to_binary(fp);
	wtz_ovlb_reader_t rd;
	if(wtz_ovlb_open(&rd, fp, NULL) != 0){ fprintf(stderr, " -- %s is not a binary overlap stream --\n", path); return 1; }
	size_t maxname = 0;
	for(uint64_t i = 0; i < rd.n_reads; i++){ const size_t l = strlen(rd.names[i]); if(l > maxname) maxname = l; }
	char *buf = (char*)malloc(2 * maxname + 256);
	static char obuf[1 << 20]; setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
	wtz_ovlb_rec_t r; unsigned long long n = 0; int st;
	while((st = wtz_ovlb_next(&rd, &r)) == 1){
		n++;
		if(summary) continue;
		size_t k = wtz_ovlb_format16(&rd, &r, buf);
		if(ncol == 17){ buf[k++] = '\t'; buf[k++] = '0'; buf[k++] = 'M'; }
		buf[k++] = '\n';
		fwrite(buf, 1, k, stdout);
	}
	if(st < 0){ fprintf(stderr, " -- truncated or corrupt binary overlap stream after %llu records --\n", n); return 1; }
	if(summary) printf("%llu reads\t%llu records\n", (unsigned long long)rd.n_reads, n);
	free(buf); wtz_ovlb_close(&rd);
	return 0;
}
