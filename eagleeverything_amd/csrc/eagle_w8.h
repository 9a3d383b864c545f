// eagle_w8.h -- W = S (V S) from int8 digit slices (eagle_w8.hip): the few types the rest of the library sees.
#ifndef EAGLE_W8_H
#define EAGLE_W8_H
#include <stdint.h>
#ifdef __cplusplus
#include <vector>
#endif

#define W8_KMAX 6         /* digits cut per operand row (48 bits below the row's scale) */
#define W8_TARGET 0.02    /* the W error the configurations are chosen for, as a fraction of (tight) budget x mean|W_kk| */
#define W8_ACCEPT 0.05    /* ... and what the finished W must meet against its own diagonal, or the call declines */

struct W8Stats {          // per operand, reduced on the device (k_w8_reduce)
    double maxd;          // max |diagonal|
    double fro2;          // sum of squares of the off-diagonal part
    double es2;           // sum_i 4^(e_i + 2) over rows with a non-zero off-diagonal part
    double wdsum;         // sum_i d2_i^2 |d_i| (second operand only)
    double phi2[W8_KMAX]; // sum_i 4^(e_i + 2) sum_l digit_p[i][l]^2
    double asym;          // max |M - M^T|
    int bad, pad;         // a non-finite entry
};
struct W8Group { int level, npairs; unsigned char p[8], q[8]; };   // digit pairs (0-based) that share one int32 accumulator
struct W8Config { int k, T; };                                     // pairs p, q <= k with p + q <= T (1-based)
struct W8Info {           // what the last eagle_dev_scan_operands_w8 did (eagle_last_w_info)
    int declined = 0;     // 0 = W came from the int8 engine; 1 non-finite, 2 asymmetric / no yardstick, 3 / 4 no configuration for product 1 / 2,
                          // 5 no workspace, 6 finished W failed the check against its own diagonal, 7 switched off / too small
    int config1 = -1, config2 = -1, k1 = 0, T1 = 0, k2 = 0, T2 = 0, pairs1 = 0, pairs2 = 0;
    double eta = 0, eta_x = 0, bound1 = 0, bound2 = 0, norm_s = 0, target = 0, mean_diag = 0, asym_term = 0;
    bool pipelined = false;   // the first product ran column block by column block under V's upload, on a guessed configuration the rule confirmed
};
#ifdef __cplusplus
int w8_config_pairs(const W8Config& c);
std::vector<W8Group> w8_groups(const W8Config& c, int maxp);
void w8_work_list(int rt0, int rt1, int ntj, int ti_rows, int tj_rows, int ui, int uj, bool upper, const std::vector<W8Group>& gs,
                  std::vector<unsigned>& out, int* maxlen_out, int tj0 = 0);
double w8_product_bound(const W8Stats& A, const W8Stats& B, const W8Config& c, long np);
#endif
#endif
