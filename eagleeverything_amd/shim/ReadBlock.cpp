// Replacement for MyPackage/Eagle/src/ReadBlock.cpp (same exported signature, ReadBlock.cpp:16-19).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
Eigen::MatrixXd ReadBlock(std::string asciifname, long start_row, long numcols, long numrows_in_block) {
    Eigen::MatrixXd M(numrows_in_block, numcols);  // column-major, as the library writes it
    eagle_check(eagle_read_block(eagle_backend_ctx(), asciifname.c_str(), start_row, numcols, numrows_in_block, M.data()));
    return M;
}
