/*
 * eagle_oracle_ingest.c -- CPU ORACLE for the marker-file ingestion next to the hot path.  TEST INFRASTRUCTURE ONLY
 * (same rules as eagle_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it).
 *
 * Plain-C restatement, line by line and token by token like the reference, of
 *   eo_getRowColumn ......... E/src/getRowColumn.cpp:20-72
 *   eo_create_ascii_text .... E/src/CreateASCIInospace.cpp:17-125   (called from createM_ASCII_rcpp.cpp:84-95)
 *   eo_create_ascii_plink ... E/src/CreateASCIInospace_PLINK.cpp:16-195 (called from createM_ASCII_rcpp.cpp:66-73)
 *   eo_createMt_ascii ....... E/src/createMt_ASCII_rcpp.cpp:14-220 (both memory branches write the same bytes)
 * (E/ = /root/reference/MyPackage/Eagle/).
 *
 * PARITY STATUS: the PLINK conversion is pinned by the reference's own data pair MyPackage/geno.ped <-> MyPackage/geno.txt
 * (the same 150 x 100 genotypes in both encodings; tests/golden/geno_150x100.{ped,txt}); the text conversion and the
 * transpose are byte shuffles with a fully specified result.  No reference outputs exist: otherwise "parity unpinned".
 *
 * Where the reference has undefined behaviour (more tokens on a line than dims[1] overruns `rowinfile`; fewer or
 * shorter lines than dims say in createMt) this restatement reports an error instead.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EO_OK 0
#define EO_ERR_OPEN (-1)
#define EO_ERR_SHORT (-2)
#define EO_ERR_NOMEM (-4)
#define EO_FALSE_TOKEN 1    /* reference returned false: genotype token is none of AA, AB, BB, missing */
#define EO_FALSE_COLUMNS 2  /* reference returned false: unequal number of columns per row */
#define EO_FALSE_ALLELES 3  /* reference returned false: more than two alleles at a locus */

/* operator>>(istream&, string&): skip whitespace, take the run of non-whitespace characters */
static const char* next_token(const char* p, const char* end, const char** tok, long* len) {
    while (p < end && isspace((unsigned char)*p)) p++;
    if (p >= end) return NULL;
    *tok = p;
    while (p < end && !isspace((unsigned char)*p)) p++;
    *len = p - *tok;
    return p;
}
static int tok_eq(const char* tok, long len, const char* s) { return (long)strlen(s) == len && memcmp(tok, s, (size_t)len) == 0; }

/* getRowColumn.cpp:41-59: rows = number of getline() successes, columns = tokens of the first line */
int eo_getRowColumn(const char* fname, long dims[2]) {
    FILE* f = fopen(fname, "r");
    if (!f) return EO_ERR_OPEN;
    char* line = NULL;
    size_t cap = 0;
    ssize_t got;
    dims[0] = dims[1] = 0;
    while ((got = getline(&line, &cap, f)) >= 0) {
        if (dims[0] == 0) {
            const char *p = line, *end = line + got, *tok;
            long len;
            while ((p = next_token(p, end, &tok, &len)) != NULL) dims[1]++;
        }
        dims[0]++;
    }
    free(line);
    fclose(f);
    return EO_OK;
}

/* CreateASCIInospace.cpp:69-122.  err_row = 1-based row of the failure, err_tok = offending token (EO_FALSE_TOKEN) or
 * err_cols = number of columns found (EO_FALSE_COLUMNS).  The output keeps the rows written before the failure. */
int eo_create_ascii_text(const char* fname, const char* asciifname, long ncols, const char* AA, const char* AB, const char* BB,
                         const char* missing, long* err_row, long* err_cols, char* err_tok, long err_tok_cap) {
    FILE* in = fopen(fname, "r");
    if (!in) return EO_ERR_OPEN;
    FILE* out = fopen(asciifname, "w");
    if (!out) { fclose(in); return EO_ERR_OPEN; }
    char* rowinfile = malloc((size_t)ncols + 2);
    if (!rowinfile) { fclose(in); fclose(out); return EO_ERR_NOMEM; }
    memset(rowinfile, '0', (size_t)ncols);
    char* line = NULL;
    size_t cap = 0;
    ssize_t got;
    long counter = 0;
    int rc = EO_OK;
    while (rc == EO_OK && (got = getline(&line, &cap, in)) >= 0) {
        const char *p = line, *end = line + got, *tok;
        long len, i = 0;
        while ((p = next_token(p, end, &tok, &len)) != NULL) {
            char c;
            if (tok_eq(tok, len, BB)) c = '2';              /* :95 */
            else if (tok_eq(tok, len, AB)) c = '1';         /* :97 */
            else if (tok_eq(tok, len, AA)) c = '0';         /* :99 */
            else if (tok_eq(tok, len, missing)) c = '1';    /* :101-103 missing genotypes become hets */
            else {                                          /* :104-116 */
                *err_row = counter + 1;
                if (err_tok && err_tok_cap > 0) { long k = len < err_tok_cap - 1 ? len : err_tok_cap - 1; memcpy(err_tok, tok, (size_t)k); err_tok[k] = 0; }
                rc = EO_FALSE_TOKEN;
                break;
            }
            if (i < ncols) rowinfile[i] = c;
            i++;
        }
        if (rc != EO_OK) break;
        if (i != ncols) { *err_row = counter + 1; *err_cols = i; rc = EO_FALSE_COLUMNS; break; }  /* :122-131 */
        rowinfile[ncols] = '\n';
        fwrite(rowinfile, 1, (size_t)ncols + 1, out);       /* :132-135 */
        counter++;
    }
    free(line);
    free(rowinfile);
    fclose(in);
    fclose(out);
    return rc;
}

/* CreateASCIInospace_PLINK.cpp:52-190.  ncols_total = dims[1] = 6 + 2 * loci.  missing_seen: the "missing alleles"
 * warning was issued (:105-116).  err_row / err_locus are 1-based like the reference's messages (:157-158). */
int eo_create_ascii_plink(const char* fname, const char* asciifname, long ncols_total, long* err_row, long* err_locus,
                          long* err_cols, int* missing_seen) {
    const long nloci = (long)((ncols_total - 6) / 2.0);      /* :20 */
    FILE* in = fopen(fname, "r");
    if (!in) return EO_ERR_OPEN;
    FILE* out = fopen(asciifname, "w");
    if (!out) { fclose(in); return EO_ERR_OPEN; }
    char* alleles0 = malloc((size_t)nloci + 1);
    char* alleles1 = malloc((size_t)nloci + 1);
    char* rowvec = malloc((size_t)(ncols_total > 6 ? ncols_total - 6 : 0) + 2);
    char* rowinfile = malloc((size_t)nloci + 2);
    if (!alleles0 || !alleles1 || !rowvec || !rowinfile) { fclose(in); fclose(out); free(alleles0); free(alleles1); free(rowvec); free(rowinfile); return EO_ERR_NOMEM; }
    memset(rowinfile, '0', (size_t)nloci);
    char* line = NULL;
    size_t cap = 0;
    ssize_t got;
    long counter = 0;
    int rc = EO_OK;
    *missing_seen = 0;
    while (rc == EO_OK && (got = getline(&line, &cap, in)) >= 0) {
        const char *end = line + got, *p = line, *tok;
        long len, numcols = 0;
        while ((p = next_token(p, end, &tok, &len)) != NULL) numcols++;           /* :60-63 */
        if (numcols != ncols_total) { *err_row = counter + 1; *err_cols = numcols; rc = EO_FALSE_COLUMNS; break; }  /* :65-74 */
        p = line;
        for (int i = 0; i <= 5; i++) p = next_token(p, end, &tok, &len);          /* :85-87 */
        for (long i = 6; i < ncols_total; i++) {                                  /* :88-90: operator>>(char&) takes ONE character */
            while (p < end && isspace((unsigned char)*p)) p++;
            rowvec[i - 6] = p < end ? *p++ : 0;
        }
        if (counter == 0) {                                                       /* :95-106 */
            for (long i = 0; i < nloci; i++) {
                const char c0 = rowvec[2 * i], c1 = rowvec[2 * i + 1];
                if (c0 == '0' || c1 == '0' || c0 == '-' || c1 == '-') { alleles0[i] = 'I'; alleles1[i] = 'I'; }
                else { alleles0[i] = c0; alleles1[i] = c1; }
            }
        }
        for (long i = 0; i < nloci && rc == EO_OK; i++) {                         /* :110-187 */
            if (rowvec[2 * i] == '0' || rowvec[2 * i + 1] == '0' || rowvec[2 * i] == '-' || rowvec[2 * i + 1] == '-') {
                *missing_seen = 1;
                rowvec[2 * i] = 'I';
                rowvec[2 * i + 1] = 'I';
            }
            for (int j = 1; j >= 0; --j) {
                const char c = rowvec[2 * i + j];
                if (c != alleles0[i] && c != alleles1[i]) {
                    if (c == 'I') {
                    } else if (alleles0[i] == 'I') {
                        alleles0[i] = c;
                    } else if (alleles1[i] == 'I') {
                        alleles1[i] = c;
                    } else if (alleles0[i] == alleles1[i]) {
                        alleles1[i] = c;
                    } else {
                        *err_row = counter + 1;
                        *err_locus = i + 1;
                        rc = EO_FALSE_ALLELES;
                        break;
                    }
                }
                if (rowvec[2 * i] == 'I' || rowvec[2 * i + 1] == 'I') rowinfile[i] = '1';
                else if (rowvec[2 * i + 1] != rowvec[2 * i]) rowinfile[i] = '1';
                else if (rowvec[2 * i] == alleles0[i]) rowinfile[i] = '0';
                else rowinfile[i] = '2';
            }
        }
        if (rc != EO_OK) break;
        rowinfile[nloci] = '\n';
        fwrite(rowinfile, 1, (size_t)nloci + 1, out);                             /* :191-192 */
        counter++;
    }
    free(line); free(alleles0); free(alleles1); free(rowvec); free(rowinfile);
    fclose(in);
    fclose(out);
    return rc;
}

/* createMt_ASCII_rcpp.cpp:86-120 (in memory) and :139-207 (column blocks): line j of the output holds character j of
 * every input line.  (c - '0') + '0' is the identity on every byte, so this is a byte transpose. */
int eo_createMt_ascii(const char* fname, const char* asciifname, long n, long L) {
    FILE* in = fopen(fname, "r");
    if (!in) return EO_ERR_OPEN;
    char* M = malloc((size_t)n * (size_t)L + 1);
    if (!M) { fclose(in); return EO_ERR_NOMEM; }
    char* line = NULL;
    size_t cap = 0;
    int rc = EO_OK;
    for (long r = 0; r < n; r++) {
        ssize_t got = getline(&line, &cap, in);
        if (got < L) { rc = EO_ERR_SHORT; break; }
        memcpy(M + (size_t)r * L, line, (size_t)L);
    }
    free(line);
    fclose(in);
    if (rc != EO_OK) { free(M); return rc; }
    FILE* out = fopen(asciifname, "w");
    if (!out) { free(M); return EO_ERR_OPEN; }
    char* row = malloc((size_t)n + 1);
    if (!row) { free(M); fclose(out); return EO_ERR_NOMEM; }
    for (long j = 0; j < L; j++) {
        for (long r = 0; r < n; r++) row[r] = M[(size_t)r * L + j];
        row[n] = '\n';
        fwrite(row, 1, (size_t)n + 1, out);
    }
    free(row);
    free(M);
    fclose(out);
    return rc;
}
