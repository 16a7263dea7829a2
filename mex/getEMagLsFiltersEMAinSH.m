function [wMlsL, wMlsR] = getEMagLsFiltersEMAinSH(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, order, fs, len, shDefinition, shFunction, chFunction)
% lib/getEMagLsFiltersEMAinSH.m:1-2 on the MI355X library (built-in getSH / getCH only)
if (nargin >= 11 && ~isequal(func2str(shFunction), 'getSH')) || (nargin >= 12 && ~isequal(func2str(chFunction), 'getCH'))
    error('eMagLS:arg', 'custom shFunction / chFunction handles are not accelerated for the EMA variants');
end
if nargin < 10 || isempty(shDefinition); shDefinition = 'real'; end
[wMlsL, wMlsR] = emagls_mex('emainsh', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
    double(micGridAziRad(:)), order, fs, len, shDefinition);
end
