/* TEST STAND-IN for MATLAB's mex.h (R2018a interleaved-complex API), just large enough to compile and drive OUR gateway
 * mex/emagls_mex.cpp in the test suite (tests/test_mex_gateway.py): argument marshalling, output allocation and error
 * forwarding are then exercised without MATLAB.  Not MATLAB's header, not shipped with the product, declares only what the
 * gateway uses. */
#ifndef EMAGLS_TEST_MEX_H
#define EMAGLS_TEST_MEX_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef size_t mwSize;
typedef struct mxArray_tag mxArray;
typedef double mxDouble;
typedef struct { double real, imag; } mxComplexDouble;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxLOGICAL_CLASS = 3, mxCELL_CLASS = 1, mxSTRUCT_CLASS = 2, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6 } mxClassID;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexPrintf(const char* fmt, ...);
int mexAtExit(void (*fn)(void));

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c);
void mxDestroyArray(mxArray* a);
mxDouble* mxGetDoubles(const mxArray* a);
mxComplexDouble* mxGetComplexDoubles(const mxArray* a);
double mxGetScalar(const mxArray* a);
mwSize mxGetM(const mxArray* a);
mwSize mxGetN(const mxArray* a);
mwSize mxGetNumberOfElements(const mxArray* a);
mwSize mxGetNumberOfDimensions(const mxArray* a);
const mwSize* mxGetDimensions(const mxArray* a);
int mxGetString(const mxArray* a, char* buf, mwSize buflen);
bool mxIsComplex(const mxArray* a);
bool mxIsDouble(const mxArray* a);
bool mxIsChar(const mxArray* a);
bool mxIsEmpty(const mxArray* a);
bool mxIsLogicalScalarTrue(const mxArray* a);
bool mxIsStruct(const mxArray* a);
bool mxIsCell(const mxArray* a);
mxArray* mxGetField(const mxArray* a, mwSize index, const char* name);   /* NULL when the struct has no such field */
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
void mxSetCell(mxArray* a, mwSize index, mxArray* value);
mxArray* mxGetCell(const mxArray* a, mwSize index);
#ifdef __cplusplus
}
#endif
#endif
