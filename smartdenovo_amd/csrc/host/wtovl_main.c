/*
 * wtovl — binary overlap records (include/wtz_ovlb.h; `wtzmo --binary-out`) back to the text the reference's consumers parse
 * (SURVEY 8f3): a consumer that stays on text - `wtclp` (load_alignments_wtclp, wtclp.c:111-180: columns 0-4, 5-9, 11 of >= 12) or
 * `wtlay` (parse_overlap_item_strgraph, wtlay.h:238-268: columns 0-12 of >= 16) - reads `wtovl x.ovlb |` instead of the file.
 *
 *   wtovl [-c 16|17] [in.ovlb | -]      16 columns: byte-identical to `cut -f1-16` of the text output (what the zmo pipeline keeps,
 *                                        smartdenovo.pl:58); 17: the same + a CIGAR column "0M" (the dmo engine's form, wtzmo.c:1243)
 *   wtovl -s [in.ovlb | -]               summary only: reads in the name table, records
 * Host text filter, no arithmetic in it, no device.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "wtz_ovlb.h"

static int usage(void){
	fputs("WTOVL: binary overlap records (wtzmo --binary-out) -> the reference's text columns\n"
	      "Usage: wtovl [-c 16|17] [-s] [file | -]\n", stdout);
	return 1;
}

int main(int argc, char **argv){
	int ncol = 16, summary = 0; const char *path = "-";
	for(int i = 1; i < argc; i++){
		if(!strcmp(argv[i], "-c") && i + 1 < argc){ ncol = atoi(argv[++i]); if(ncol != 16 && ncol != 17) return usage(); }
		else if(!strcmp(argv[i], "-s")) summary = 1;
		else if(!strcmp(argv[i], "-h")) return usage();
		else path = argv[i];
	}
	FILE *fp = strcmp(path, "-") ? fopen(path, "rb") : stdin;
	if(!fp){ fprintf(stderr, " -- Cannot open %s --\n", path); return 1; }
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
