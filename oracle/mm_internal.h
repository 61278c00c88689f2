/* mm_internal.h -- CPU ORACLE (test infrastructure): helpers shared by the oracle's own files. */
#ifndef MM_INTERNAL_H
#define MM_INTERNAL_H
#include "mm_oracle.h"

#define ORC_PARENT_UNSET   (-1)
#define ORC_PARENT_TMP_PRI (-2)
#define ORC_SEED_LONG_JOIN (1ULL<<40)
#define ORC_SEED_IGNORE    (1ULL<<41)
#define ORC_SEED_TANDEM    (1ULL<<42)

unsigned char orc_nt4(unsigned char c);
void orc_reg_set_coor(orc_reg_t *r, int32_t qlen, const orc128_t *a);
void orc_sync_regs(int n_regs, orc_reg_t *regs);
int orc_squeeze_a(int n_regs, orc_reg_t *regs, orc128_t *a);
void orc_set_parent(float mask_level, int n, orc_reg_t *r, int sub_diff);
void orc_select_sub(float pri_ratio, int min_diff, int best_n, int *n_, orc_reg_t *r);
void orc_filter_regs(const orc_opt_t *opt, int qlen, int *n_regs, orc_reg_t *regs);
#endif
