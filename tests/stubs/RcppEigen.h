// TEST INFRASTRUCTURE ONLY: see Rcpp.h in this directory.
#include "Rcpp.h"
