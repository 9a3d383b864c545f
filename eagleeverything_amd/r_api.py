"""Mirror of the thin R wrappers around the hot calls (argument marshalling is part of parity).

  calculateMMt ........... E/R/calculateMMt.R:3-30
  calcMMt ................ E/R/calcMMt.R:1-15       (.calcMMt)
  calculate_a_and_vara ... E/R/calculate_a_and_vara.R:1-34
  find_qtl ............... E/R/find_qtl.R:1-84      (.find_qtl; host algebra from host_model)
  extract_geno, constructX E/R/extract_geno.R:1-19, E/R/constructX.R:1-24

`geno` is the reference's list {asciifileM, asciifileMt, dim_of_ascii_M = (n, L)} (E/R/ReadMarker.R:306-307).
selected_loci follow R: 1-based, NA = numpy.nan.  The "-1 only if no NA anywhere" rule
(calculateMMt.R:24, calculate_a_and_vara.R:23) is reproduced literally.
"""
import os

import numpy as np

from . import host_model, rcpp_api


def _shift_if_no_na(selected_loci):
    s = np.atleast_1d(np.asarray(selected_loci, dtype=np.float64))
    if not np.any(np.isnan(s)):
        s = s - 1
    return s


def calculateMMt(geno, availmemGb, ncpu, selected_loci=np.nan, dim_of_ascii_M=None, quiet=True, message=None, device=0):
    if not os.path.exists(geno):  # calculateMMt.R:19-23
        if message:
            message(" Error: The binary packed file %s cannot be found.\n" % geno)
            message(" calculateMMt has terminated with errors.")
        return None
    return rcpp_api.calculateMMt_rcpp(f_name_ascii=geno, selected_loci=_shift_if_no_na(selected_loci),
                                      max_memory_in_Gbytes=availmemGb, num_cores=ncpu, dims=dim_of_ascii_M, quiet=quiet,
                                      message=message, device=device)


def calcMMt(geno, availmemGb, ncpu, selected_loci, quiet, device=0, message=None):
    MMt = calculateMMt(geno=geno["asciifileM"], availmemGb=availmemGb, ncpu=ncpu,
                       dim_of_ascii_M=geno["dim_of_ascii_M"], selected_loci=selected_loci, quiet=quiet, message=message,
                       device=device)
    if MMt is None:
        return None
    # MMt/max(MMt) + diag(0.95): evaluated on the device from the result still held in HBM (calcMMt.R:13)
    out, _ = rcpp_api.last_mmt_normalised(MMt.shape[0], device=device)
    return out


def calculate_a_and_vara(geno, maxmemGb=8, selectedloci=np.nan, invMMtsqrt=None, transformed_a=None,
                         transformed_vara=None, quiet=True, message=None, device=0):
    fnameMt = geno["asciifileMt"]
    dimsMt = (geno["dim_of_ascii_M"][1], geno["dim_of_ascii_M"][0])  # calculate_a_and_vara.R:21
    return rcpp_api.calculate_a_and_vara_rcpp(f_name_ascii=fnameMt, selected_loci=_shift_if_no_na(selectedloci),
                                              inv_MMt_sqrt=invMMtsqrt, dim_reduced_vara=transformed_vara,
                                              max_memory_in_Gbytes=maxmemGb, dims=dimsMt, a=transformed_a, quiet=quiet,
                                              message=message, device=device)


def extract_geno(fnameM, colnum, availmemGb=8, dim_of_ascii_M=None, device=0):
    """E/R/extract_geno.R:1-19 (colnum is 1-based; the C++ side is 0-based)."""
    return rcpp_api.extract_geno_rcpp(f_name_ascii=fnameM, max_memory_in_Gbytes=availmemGb, selected_locus=colnum - 1,
                                      dims=dim_of_ascii_M, device=device)


def constructX(fnameM, currentX, loci_indx, availmemGb=8, dim_of_ascii_M=None, device=0):
    """E/R/constructX.R:1-24 (column names are the R caller's business)."""
    if loci_indx is None or (isinstance(loci_indx, float) and np.isnan(loci_indx)):
        return currentX
    g = extract_geno(fnameM, int(loci_indx), availmemGb, dim_of_ascii_M, device=device)
    return np.column_stack([currentX, g.astype(np.float64)])


def find_qtl(geno, availmemGb, selected_loci, MMt, invMMt, best_ve, best_vg, currentX, ncpu, quiet, trait, ngpu=1,
             device=0, return_stats=False):
    """E/R/find_qtl.R:1-84.  Host algebra (H, P, MMt^{+-1/2}, a_hat, Var a_hat) on host LAPACK, the genome scan and
    the arg-max on the GPU.  Returns the 1-based column of the selected marker."""
    H = host_model.calculateH(MMt, best_ve, best_vg)
    P = host_model.calculateP(H, currentX)
    sq = host_model.calculateMMt_sqrt_and_sqrtinv(MMt, checkres=not quiet)
    hat_a = host_model.calculate_reduced_a(best_vg, P, sq["sqrt_MMt"], trait)
    var_hat_a = host_model.calculate_reduced_vara(currentX, best_ve, best_vg, invMMt, sq["sqrt_MMt"])
    a_and_vara = calculate_a_and_vara(geno=geno, maxmemGb=availmemGb, selectedloci=selected_loci,
                                      invMMtsqrt=sq["inverse_sqrt_MMt"], transformed_a=hat_a,
                                      transformed_vara=var_hat_a, quiet=quiet, device=device)
    indx, tsqmax, near = rcpp_api.last_scan_argmax(device=device)  # find_qtl.R:71-83 on the device
    if return_stats:
        return indx, {"tsqmax": tsqmax, "near_ties": near, "a": a_and_vara["a"], "vara": a_and_vara["vara"]}
    return indx


def create_ascii(file_genotype, type="text", AA=None, AB=None, BB=None, availmemGb=8, dim_of_ascii_M=None, quiet=True,
                 missing=None, outdir=None, message=None, device=0):
    """E/R/create_ascii.R:1-62 -> True / False; writes <outdir>/M.ascii and <outdir>/Mt.ascii (R: tempdir())."""
    outdir = outdir or os.path.dirname(os.path.abspath(file_genotype))
    asciiMfile, asciiMtfile = os.path.join(outdir, "M.ascii"), os.path.join(outdir, "Mt.ascii")
    dims = [int(dim_of_ascii_M[0]), int(dim_of_ascii_M[1])]
    if type == "text":
        missing = "NA" if missing is None else str(missing)                       # :30-34
        if not rcpp_api.createM_ASCII_rcpp(file_genotype, asciiMfile, type, AA, AB, BB, availmemGb, dims, quiet, message, missing,
                                           device=device):
            return False
        rcpp_api.createMt_ASCII_rcpp(asciiMfile, asciiMtfile, type, availmemGb, dims, quiet, message, device=device)
    else:
        ncol = dims[1]
        dims[1] = 2 * dims[1] + 6                                                   # :46-47 columns of a PLINK ped file
        if not rcpp_api.createM_ASCII_rcpp(file_genotype, asciiMfile, type, "-9", "-9", "-9", availmemGb, dims, quiet, message, "NA",
                                           device=device):
            return False
        dims[1] = ncol                                                              # :54
        rcpp_api.createMt_ASCII_rcpp(asciiMfile, asciiMtfile, type, availmemGb, dims, quiet, message, device=device)
    return True


def ReadMarker(filename=None, type="text", missing=None, AA=None, AB=None, BB=None, availmemGb=16, quiet=True, outdir=None,
               message=None, device=0):
    """E/R/ReadMarker.R:194-318 -> geno dict {asciifileM, asciifileMt, dim_of_ascii_M} or None (the R list / NULL)."""
    say = message or (lambda s: None)
    if type not in ("text", "PLINK"):                                               # :206-215
        say(' type must be set to "text" or "PLINK". \n')
        say(" ReadMarker has terminated with errors")
        return None
    if filename is None or not os.path.exists(filename):                            # :222-231, check_inputs.R
        say(" The %s file %s could not be found. " % ("PLINK ped" if type == "PLINK" else "marker", filename))
        say(" ReadMarker has terminated with errors ")
        return None
    genofile = os.path.abspath(filename)
    outdir = outdir or os.path.dirname(genofile)
    if type == "PLINK":
        dims = rcpp_api.getRowColumn(genofile, device=device)                       # :234-235
        dims[1] = (dims[1] - 6) // 2
        ok = create_ascii(genofile, type=type, availmemGb=availmemGb, dim_of_ascii_M=dims, quiet=quiet, outdir=outdir,
                          message=message, device=device)
    else:
        if AA is None or BB is None:                                                # :262-268
            say("Error: The function parameters AA and BB must be assigned a numeric or character value since a text file is being assumed. \n")
            say(" ReadMarker has terminated with errors")
            return None
        if AB is None:
            AB = "NA"                                                               # :271-272 no hets
        say(" Getting number of individuals and snp from file ... ")
        dims = rcpp_api.getRowColumn(genofile, device=device)                       # :283
        say(" Beginning creation of reformatted file ... ")
        ok = create_ascii(genofile, type=type, AA=str(AA), AB=str(AB), BB=str(BB), availmemGb=availmemGb, dim_of_ascii_M=dims,
                          quiet=quiet, missing=missing, outdir=outdir, message=message, device=device)
    if not ok:
        return None
    return {"asciifileM": os.path.join(outdir, "M.ascii"), "asciifileMt": os.path.join(outdir, "Mt.ascii"), "dim_of_ascii_M": dims}
