// TEST INFRASTRUCTURE ONLY -- not part of the product, not a build of the reference, never linked or run.
// A container without R cannot compile eagleeverything_amd/shim/*.cpp; this header declares just enough of the few
// Rcpp / Eigen types those eight translation units touch for `g++ -fsyntax-only` to type-check them against
// include/eagle_hip.h (tests/test_abi.py::test_shims_typecheck).  Signatures follow the reference's exported
// prototypes (E/src/RcppExports.cpp:9-151); nothing here has behaviour.
#ifndef EAGLE_TEST_STUB_RCPP_H
#define EAGLE_TEST_STUB_RCPP_H
#include <cstdlib>
#include <string>
#include <vector>
namespace Rcpp {
struct CharacterVector {};
struct NumericVector { const double* begin() const; long size() const; };
struct Function { void operator()(const char*) const; };
template <class T> T as(const CharacterVector&);
[[noreturn]] void stop(const std::string&);
template <class T> struct NamedValue {};
struct Named { explicit Named(const char*); template <class T> NamedValue<T> operator=(const T&) const; };
struct List { template <class... A> static List create(const A&...); };
}  // namespace Rcpp
namespace Eigen {
template <class T> struct StubMatrix {
    StubMatrix(long rows, long cols);
    explicit StubMatrix(long rows);
    T* data();
    static StubMatrix Zero(long rows, long cols);
};
typedef StubMatrix<double> MatrixXd;
typedef StubMatrix<double> VectorXd;
typedef StubMatrix<int> VectorXi;
template <class M> struct Map { const double* data() const; };
}  // namespace Eigen
#endif
