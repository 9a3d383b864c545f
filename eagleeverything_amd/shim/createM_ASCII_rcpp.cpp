// Replacement for MyPackage/Eagle/src/createM_ASCII_rcpp.cpp (same exported signature, :18-27); CreateASCIInospace.cpp and
// CreateASCIInospace_PLINK.cpp are no longer compiled (their work happens inside eagle_create_M_ascii).
// [[Rcpp::depends(RcppEigen)]]
#include <RcppEigen.h>

#include "eagle_backend.h"

// [[Rcpp::export]]
bool createM_ASCII_rcpp(Rcpp::CharacterVector f_name, Rcpp::CharacterVector f_name_ascii, Rcpp::CharacterVector type, std::string AA,
                        std::string AB, std::string BB, double max_memory_in_Gbytes, std::vector<long> dims, bool quiet,
                        Rcpp::Function message, std::string missing) {
    EagleMessageScope scope(message);
    const std::string in = Rcpp::as<std::string>(f_name), out = Rcpp::as<std::string>(f_name_ascii), ftype = Rcpp::as<std::string>(type);
    const long d[2] = {dims[0], dims[1]};
    const int rc = eagle_create_M_ascii(eagle_backend_ctx(), in.c_str(), out.c_str(), ftype.c_str(), AA.c_str(), AB.c_str(), BB.c_str(),
                                        max_memory_in_Gbytes, d, quiet, missing.c_str());
    eagle_check(rc);
    return rc == EAGLE_OK;  // EAGLE_SOFT_SENTINEL: the messages were sent, the reference returns false here
}
