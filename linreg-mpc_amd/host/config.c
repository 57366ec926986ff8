/* config.c -- header of the input file.  Format (README of the reference, "Input format"):
 *   n d P | endpoint of party 1 | endpoint of party 2 | P x (endpoint first_column) | data ...
 * The stream is left positioned at the first token after the header. */
#include "config.h"

#include <stdlib.h>
#include <string.h>

/* next whitespace-separated token, or NULL at end of input */
static char *next_token(FILE *f) {
    char buf[512];
    return fscanf(f, "%511s", buf) == 1 ? strdup(buf) : NULL;
}

static int next_size(FILE *f, size_t *out) {
    char *tok = next_token(f), *end = NULL;
    if (!tok) return -1;
    unsigned long long v = strtoull(tok, &end, 10);
    int bad = (*tok == '\0' || *end != '\0');
    free(tok);
    if (bad) return -1;
    *out = (size_t)v;
    return 0;
}

static int fail(config **cfg, const char *what, int party) {
    if (party > 0) fprintf(stderr, "Error reading %s of party %d: Invalid input\n", what, party);
    else fprintf(stderr, "Error reading %s: Invalid input\n", what);
    config_destroy(cfg);
    return 1;
}

int config_new(config **cfg, const char *filename) {
    size_t providers = 0;
    config *c = *cfg = calloc(1, sizeof(config));
    if (!c) return 1;
    if (!(c->input = fopen(filename, "r"))) {
        perror("fopen");
        config_destroy(cfg);
        return 1;
    }
    if (next_size(c->input, &c->n) || next_size(c->input, &c->d) || next_size(c->input, &providers)) return fail(cfg, "config", 0);
    c->num_parties = (int)providers + 2;          /* CSP and Evaluator come first */
    c->endpoint = calloc((size_t)c->num_parties, sizeof(char *));
    c->index_owned = calloc((size_t)c->num_parties, sizeof(ssize_t));
    if (!c->endpoint || !c->index_owned) { config_destroy(cfg); return 1; }
    for (int k = 0; k < c->num_parties; k++) {
        if (!(c->endpoint[k] = next_token(c->input))) return fail(cfg, "endpoint", k + 1);
        c->index_owned[k] = -1;                   /* parties 1 and 2 own no columns */
        if (k >= 2) {
            size_t first;
            if (next_size(c->input, &first)) return fail(cfg, "index", k + 1);
            c->index_owned[k] = (ssize_t)first;
        }
    }
    return 0;
}

void config_destroy(config **cfg) {
    config *c = cfg ? *cfg : NULL;
    if (!c) return;
    if (c->input) fclose(c->input);
    for (int k = 0; c->endpoint && k < c->num_parties; k++) free(c->endpoint[k]);
    free(c->endpoint);
    free(c->index_owned);
    free(c);
    *cfg = NULL;
}

/* the data provider (0-based party index) holding column `row`; row == d is the target vector,
 * held by the last provider */
int config_owner(const config *c, size_t row) {
    int owner = 2;
    for (int k = 3; k < c->num_parties; k++)
        if (c->index_owned[k] <= (ssize_t)row) owner = k;
    return owner;
}
