#include "config.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>

int config_new(config **out, const char *filename) {
    config *c = calloc(1, sizeof *c);
    if (!c) return 1;
    *out = c;
    c->input = fopen(filename, "r");
    if (!c->input) { fprintf(stderr, "fopen: %s\n", strerror(errno)); goto fail; }
    if (fscanf(c->input, "%zu %zu %d", &c->n, &c->d, &c->num_parties) != 3) { fprintf(stderr, "Error reading config: Invalid input\n"); goto fail; }
    c->num_parties += 2;
    c->endpoint = calloc((size_t)c->num_parties, sizeof(char *));
    c->index_owned = calloc((size_t)c->num_parties, sizeof(ssize_t));
    for (int i = 0; i < c->num_parties; i++) {
        char buf[512];
        if (fscanf(c->input, "%511s", buf) != 1) { fprintf(stderr, "Error reading endpoint for party %d: Invalid input\n", i + 1); goto fail; }
        c->endpoint[i] = strdup(buf);
        if (i < 2) c->index_owned[i] = -1;
        else if (fscanf(c->input, "%zd", &c->index_owned[i]) != 1) { fprintf(stderr, "Error reading index of party %d: Invalid input\n", i + 1); goto fail; }
    }
    return 0;
fail:
    config_destroy(out);
    return 1;
}

void config_destroy(config **cc) {
    if (!cc || !*cc) return;
    config *c = *cc;
    if (c->input) fclose(c->input);
    if (c->endpoint) { for (int i = 0; i < c->num_parties; i++) free(c->endpoint[i]); free(c->endpoint); }
    free(c->index_owned);
    free(c);
    *cc = 0;
}

int config_owner(const config *c, size_t row) {
    int party = 0;
    for (; party + 1 < c->num_parties && c->index_owned[party + 1] <= (ssize_t)row; party++) {}
    return party;
}
