// Replacement for MyPackage/Eagle/src/calculateMMt_rcpp.cpp (same exported signature, :19-22).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
Eigen::MatrixXd calculateMMt_rcpp(Rcpp::CharacterVector f_name_ascii, double max_memory_in_Gbytes, int num_cores,
                                  Rcpp::NumericVector selected_loci, std::vector<long> dims, bool quiet,
                                  Rcpp::Function message) {
    EagleMessageScope scope(message);
    std::string path = Rcpp::as<std::string>(f_name_ascii);
    Eigen::MatrixXd MMt(dims[0], dims[0]);
    const long d[2] = {dims[0], dims[1]};
    // selected_loci goes through as raw doubles: NA_real_ is a NaN, and the "element 0 is NA => no masking"
    // rule (calculateMMt_rcpp.cpp:88) lives inside the library.
    eagle_check(eagle_calculateMMt(eagle_backend_ctx(), path.c_str(), max_memory_in_Gbytes, num_cores, selected_loci.begin(),
                                   selected_loci.size(), d, quiet, MMt.data()));
    return MMt;
}
