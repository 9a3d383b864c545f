// Replacement for MyPackage/Eagle/src/calculate_a_and_vara_rcpp.cpp (same exported signature, :22-30).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
Rcpp::List calculate_a_and_vara_rcpp(Rcpp::CharacterVector f_name_ascii, Rcpp::NumericVector selected_loci,
                                     Eigen::Map<Eigen::MatrixXd> inv_MMt_sqrt, Eigen::Map<Eigen::MatrixXd> dim_reduced_vara,
                                     double max_memory_in_Gbytes, std::vector<long> dims, Eigen::VectorXd a, bool quiet,
                                     Rcpp::Function message) {
    EagleMessageScope scope(message);
    std::string path = Rcpp::as<std::string>(f_name_ascii);
    Eigen::MatrixXd ans(dims[0], 1), var_ans(dims[0], 1);
    const long d[2] = {dims[0], dims[1]};
    int rc = eagle_calculate_a_and_vara(eagle_backend_ctx(), path.c_str(), selected_loci.begin(), selected_loci.size(),
                                        inv_MMt_sqrt.data(), dim_reduced_vara.data(), max_memory_in_Gbytes, d, a.data(),
                                        quiet, ans.data(), var_ans.data());
    eagle_check(rc);
    if (rc == EAGLE_SOFT_SENTINEL)  // calculate_a_and_vara_rcpp.cpp:141-142
        return Rcpp::List::create(Rcpp::Named("a") = 0, Rcpp::Named("vara") = 0);
    return Rcpp::List::create(Rcpp::Named("a") = ans, Rcpp::Named("vara") = var_ans);
}
