// Replacement for MyPackage/Eagle/src/calculate_reduced_a_rcpp.cpp (same exported signature, :20-27).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
Eigen::MatrixXd calculate_reduced_a_rcpp(Rcpp::CharacterVector f_name_ascii, double varG, Eigen::Map<Eigen::MatrixXd> P,
                                         Eigen::Map<Eigen::MatrixXd> y, double max_memory_in_Gbytes, std::vector<long> dims,
                                         Rcpp::NumericVector selected_loci, bool quiet, Rcpp::Function message) {
    EagleMessageScope scope(message);
    std::string path = Rcpp::as<std::string>(f_name_ascii);
    Eigen::MatrixXd ar(dims[1], 1);
    const long d[2] = {dims[0], dims[1]};
    int rc = eagle_calculate_reduced_a(eagle_backend_ctx(), path.c_str(), varG, P.data(), y.data(), max_memory_in_Gbytes, d,
                                       selected_loci.begin(), selected_loci.size(), quiet, ar.data());
    eagle_check(rc);
    if (rc == EAGLE_SOFT_SENTINEL) return Eigen::MatrixXd::Zero(1, 1);  // calculate_reduced_a_rcpp.cpp:49,101
    return ar;
}
