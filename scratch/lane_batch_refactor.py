"""One-off refactor: add a batch (design) dimension to the pre/post-sweep kernels.
Every listed kernel gets a trailing `size_t bstride` parameter and offsets its pointer parameters by
blockIdx.z * bstride; every launch gets grid.z = batch_ctx().n and passes batch_ctx().stride."""
import re, sys, os
CS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emagls_amd", "csrc")
KERNELS = {
    "sh_basis.hip": ["sh_coeff_kernel", "sh_basis_kernel", "transpose_conj_kernel", "zero_fill_kernel"],
    "modal.hip": ["modal_bn_kernel"],
    "fft.hip": ["twiddle_kernel", "hrir_dirsum_kernel", "grpdelay_median_kernel", "hrir_fft_kernel", "filter_epilogue_kernel"],
    "gram_chol.hip": ["gram_mfma_kernel", "gram_reduce_kernel", "chol_diag_kernel", "chol_panel_kernel", "chol_update_kernel",
                      "rinv_diag_kernel", "qform_kernel", "tn_kernel", "small_gemm_kernel"],
    "dspace.hip": ["qt_kernel", "dspace_g_kernel", "dspace_yri_kernel", "cond_flag_kernel", "yri_accurate_kernel"],
    "sweep.hip": ["hq_kernel", "widen_kernel", "conj_copy_kernel"],
}

def match(s, i, o, c):
    d = 0
    while True:
        if s[i] == o: d += 1
        elif s[i] == c:
            d -= 1
            if d == 0: return i
        i += 1

def split_top(a):
    out, d, cur = [], 0, ""
    for ch in a:
        if ch in "(<[": d += 1
        if ch in ")>]": d -= 1
        if ch == "," and d == 0:
            out.append(cur); cur = ""
        else: cur += ch
    out.append(cur)
    return out

for fn, ks in KERNELS.items():
    p = os.path.join(CS, fn); s = open(p).read()
    for k in ks:
        # ---- definition
        m = re.search(r"__global__[^;{]*?\b%s\(" % k, s)
        assert m, k
        i0 = m.end() - 1; i1 = match(s, i0, "(", ")")
        params = s[i0 + 1:i1]
        names = []
        for prm in split_top(params):
            if "*" in prm:
                names.append(re.findall(r"(\w+)\s*$", prm.strip())[0])
        s = s[:i1] + ", size_t bstride" + s[i1:]
        b0 = s.index("{", i1)
        ins = "\n    " + " ".join("%s = boff(%s, bstride);" % (n, n) for n in names)
        s = s[:b0 + 1] + ins + s[b0 + 1:]
        # ---- launches
        pos = 0
        while True:
            m = re.search(r"\b%s(<[^<>]*(?:<[^<>]*>)?[^<>]*>)?<<<" % k, s[pos:])
            if not m: break
            g0 = pos + m.end()
            # launch config up to >>>
            g1 = s.index(">>>", g0)
            cfg = split_top(s[g0:g1])
            cfg[0] = "bgrid(" + cfg[0].strip() + ")"
            newcfg = ",".join([cfg[0]] + cfg[1:])
            s = s[:g0] + newcfg + s[g1:]
            g1 = g0 + len(newcfg)
            a0 = g1 + 3
            assert s[a0] == "(", s[a0:a0+20]
            a1 = match(s, a0, "(", ")")
            s = s[:a1] + ", batch_ctx().stride" + s[a1:]
            pos = a1
    open(p, "w").write(s)
    print("ok", fn)
