// Replacement for MyPackage/Eagle/src/extract_geno_rcpp.cpp (same exported signature, :16-19).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
Eigen::VectorXi extract_geno_rcpp(Rcpp::CharacterVector f_name_ascii, double max_memory_in_Gbytes, long selected_locus,
                                  std::vector<long> dims) {
    std::string path = Rcpp::as<std::string>(f_name_ascii);
    Eigen::VectorXi column_of_genos(dims[0]);
    const long d[2] = {dims[0], dims[1]};
    eagle_check(eagle_extract_geno(eagle_backend_ctx(), path.c_str(), max_memory_in_Gbytes, selected_locus, d, column_of_genos.data()));
    return column_of_genos;
}
