/* config.h -- input-file header (counterpart of the reference's src/config.c:24-44):
 *   n d P / CSP endpoint / Evaluator endpoint / P lines "endpoint first_column" / then the data */
#ifndef LINREG_CONFIG_H
#define LINREG_CONFIG_H
#include <stdio.h>
#include <sys/types.h>
typedef struct {
    int party, num_parties;     /* num_parties includes CSP and Evaluator */
    char **endpoint;
    ssize_t *index_owned;       /* -1 for parties 1 and 2 */
    size_t n, d;
    FILE *input;                /* positioned at the matrix */
} config;
int config_new(config **c, const char *filename);
void config_destroy(config **c);
/* 0-based party index owning row `row` (row d = target -> last DP), src/phase1.c:25-33 */
int config_owner(const config *c, size_t row);
#endif
