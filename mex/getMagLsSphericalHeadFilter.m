function [wShf, W_Shf] = getMagLsSphericalHeadFilter(micRadius, order, fs, len)
% lib/getMagLsSphericalHeadFilter.m:1 on the MI355X library
[wShf, W_Shf] = emagls_mex('shf', micRadius, order, fs, len);
end
