// Replacement for MyPackage/Eagle/src/getRowColumn.cpp (same exported signature, :20).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
std::vector<long> getRowColumn(std::string fname) {
    long d[2] = {0, 0};
    eagle_check(eagle_get_row_column(eagle_backend_ctx(), fname.c_str(), d));
    return std::vector<long>{d[0], d[1]};
}
