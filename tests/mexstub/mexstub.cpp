// The stand-in's implementation (see mex.h in this directory) and a C API for the Python test: build arrays from NumPy
// buffers in MATLAB's column-major layout, call mexFunction, read the outputs back.  mexErrMsgIdAndTxt throws, stub_call
// catches and returns the message, like MATLAB turning it into an error().
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    bool cplx = false, logical_value = false;
    std::vector<mwSize> dims;
    std::vector<double> data;   // interleaved (re, im) when cplx
    std::string text;
    std::vector<std::string> fields;     // struct arrays: element i, field f at cells[i * fields.size() + f]
    std::vector<mxArray_tag*> cells;     // cell arrays and struct arrays own their elements
    ~mxArray_tag() { for (auto* c : cells) delete c; }
    mwSize numel() const { mwSize n = 1; for (mwSize d : dims) n *= d; return n; }
};
namespace {
struct MexError : std::runtime_error { std::string id; MexError(const std::string& i, const std::string& m) : std::runtime_error(m), id(i) {} };
void (*g_at_exit)(void) = nullptr;
}
extern "C" {
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw MexError(id ? id : "", buf);
}
int mexPrintf(const char* fmt, ...) { va_list ap; va_start(ap, fmt); const int n = vprintf(fmt, ap); va_end(ap); return n; }
int mexAtExit(void (*fn)(void)) { g_at_exit = fn; return 0; }
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c) {
    mxArray* a = new mxArray_tag;
    a->cls = cls; a->cplx = c == mxCOMPLEX; a->dims.assign(dims, dims + ndim);
    if (a->dims.size() < 2) a->dims.resize(2, 1);
    a->data.assign(a->numel() * (a->cplx ? 2 : 1), 0.0);
    return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { const mwSize d[2] = {m, n}; return mxCreateNumericArray(2, d, mxDOUBLE_CLASS, c); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->data[0] = v; return a; }
void mxDestroyArray(mxArray* a) { delete a; }
mxDouble* mxGetDoubles(const mxArray* a) { return (a->cls == mxDOUBLE_CLASS && !a->cplx) ? const_cast<double*>(a->data.data()) : nullptr; }
mxComplexDouble* mxGetComplexDoubles(const mxArray* a) { return (a->cls == mxDOUBLE_CLASS && a->cplx) ? reinterpret_cast<mxComplexDouble*>(const_cast<double*>(a->data.data())) : nullptr; }
double mxGetScalar(const mxArray* a) { return a->cls == mxLOGICAL_CLASS ? (a->logical_value ? 1.0 : 0.0) : (a->data.empty() ? 0.0 : a->data[0]); }
mwSize mxGetM(const mxArray* a) { return a->dims[0]; }
mwSize mxGetN(const mxArray* a) { mwSize n = 1; for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i]; return n; }
mwSize mxGetNumberOfElements(const mxArray* a) { return a->numel(); }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->dims.size(); }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
    const size_t n = a->text.size() < buflen - 1 ? a->text.size() : buflen - 1;
    std::memcpy(buf, a->text.data(), n); buf[n] = 0;
    return a->text.size() >= buflen;
}
bool mxIsComplex(const mxArray* a) { return a->cplx; }
bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
bool mxIsEmpty(const mxArray* a) { return a->numel() == 0; }
bool mxIsLogicalScalarTrue(const mxArray* a) { return a->cls == mxLOGICAL_CLASS && a->numel() == 1 && a->logical_value; }
bool mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
bool mxIsCell(const mxArray* a) { return a->cls == mxCELL_CLASS; }
mxArray* mxGetField(const mxArray* a, mwSize index, const char* name) {
    if (a->cls != mxSTRUCT_CLASS || index >= a->numel()) return nullptr;
    for (size_t f = 0; f < a->fields.size(); ++f) if (a->fields[f] == name) return a->cells[index * a->fields.size() + f];
    return nullptr;
}
mxArray* mxCreateCellMatrix(mwSize m, mwSize n) {
    mxArray* a = new mxArray_tag;
    a->cls = mxCELL_CLASS; a->dims = {m, n}; a->cells.assign(m * n, nullptr);
    return a;
}
void mxSetCell(mxArray* a, mwSize index, mxArray* value) { if (a->cls == mxCELL_CLASS && index < a->cells.size()) { delete a->cells[index]; a->cells[index] = value; } }
mxArray* mxGetCell(const mxArray* a, mwSize index) { return (a->cls == mxCELL_CLASS && index < a->cells.size()) ? a->cells[index] : nullptr; }

// ---- the test's side ----
void* stub_array(int ndim, const size_t* dims, const double* data, int is_complex) {
    std::vector<mwSize> d(dims, dims + ndim);
    mxArray* a = mxCreateNumericArray((mwSize)ndim, d.data(), mxDOUBLE_CLASS, is_complex ? mxCOMPLEX : mxREAL);
    if (data) std::memcpy(a->data.data(), data, sizeof(double) * a->data.size());
    return a;
}
void* stub_string(const char* s) {
    mxArray* a = new mxArray_tag;
    a->cls = mxCHAR_CLASS; a->text = s; a->dims = {1, (mwSize)a->text.size()};
    return a;
}
void* stub_logical(int v) {
    mxArray* a = new mxArray_tag;
    a->cls = mxLOGICAL_CLASS; a->logical_value = v != 0; a->dims = {1, 1};
    return a;
}
// a 1 x n struct array with the given field names; stub_struct_set hands a value over (the struct owns it; NULL = [])
void* stub_struct(size_t n, int nfields, const char** names) {
    mxArray* a = new mxArray_tag;
    a->cls = mxSTRUCT_CLASS; a->dims = {1, (mwSize)n};
    for (int f = 0; f < nfields; ++f) a->fields.push_back(names[f]);
    a->cells.assign(n * (size_t)nfields, nullptr);
    return a;
}
void stub_struct_set(void* s, size_t index, int field, void* value) {
    mxArray* a = static_cast<mxArray*>(s);
    mxArray*& slot = a->cells[index * a->fields.size() + (size_t)field];
    delete slot;
    slot = static_cast<mxArray*>(value);
}
int stub_is_cell(const void* a) { return static_cast<const mxArray*>(a)->cls == mxCELL_CLASS; }
void* stub_cell_get(const void* a, size_t index) { return mxGetCell(static_cast<const mxArray*>(a), index); }
void stub_free(void* a) { delete static_cast<mxArray*>(a); }
int stub_ndim(const void* a) { return (int)static_cast<const mxArray*>(a)->dims.size(); }
void stub_dims(const void* a, size_t* out) { const mxArray* x = static_cast<const mxArray*>(a); for (size_t i = 0; i < x->dims.size(); ++i) out[i] = x->dims[i]; }
int stub_is_complex(const void* a) { return static_cast<const mxArray*>(a)->cplx; }
const double* stub_data(const void* a) { return static_cast<const mxArray*>(a)->data.data(); }
// 0 = returned normally; 1 = the gateway raised an error (message in err)
int stub_call(int nlhs, void** plhs, int nrhs, void** prhs, char* err, size_t errlen) {
    std::vector<mxArray*> out((size_t)(nlhs > 1 ? nlhs : 1), nullptr);
    try {
        mexFunction(nlhs, out.data(), nrhs, const_cast<const mxArray**>(reinterpret_cast<mxArray**>(prhs)));
    } catch (const MexError& e) {
        snprintf(err, errlen, "%s: %s", e.id.c_str(), e.what());
        for (mxArray* a : out) delete a;
        return 1;
    }
    for (int i = 0; i < nlhs; ++i) plhs[i] = out[(size_t)i];
    if (nlhs == 0) delete out[0];   // (MATLAB's `ans`)
    return 0;
}
void stub_at_exit(void) { if (g_at_exit) g_at_exit(); }
}
