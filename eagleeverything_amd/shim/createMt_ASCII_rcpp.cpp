// Replacement for MyPackage/Eagle/src/createMt_ASCII_rcpp.cpp (same exported signature, :14-19).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
void createMt_ASCII_rcpp(Rcpp::CharacterVector f_name, Rcpp::CharacterVector f_name_ascii, Rcpp::CharacterVector type,
                         double max_memory_in_Gbytes, std::vector<long> dims, bool quiet, Rcpp::Function message) {
    EagleMessageScope scope(message);
    const std::string in = Rcpp::as<std::string>(f_name), out = Rcpp::as<std::string>(f_name_ascii), ftype = Rcpp::as<std::string>(type);
    const long d[2] = {dims[0], dims[1]};
    eagle_check(eagle_create_Mt_ascii(eagle_backend_ctx(), in.c_str(), out.c_str(), ftype.c_str(), max_memory_in_Gbytes, d, quiet));
}
