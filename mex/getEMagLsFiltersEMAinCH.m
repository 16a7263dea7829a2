function [wMlsL, wMlsR] = getEMagLsFiltersEMAinCH(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, order, fs, len, shDefinition, shFunction, chFunction)
if nargin >= 12 && ~isequal(func2str(chFunction), 'getCH'); error('eMagLS:arg', 'only the built-in getCH is accelerated'); end
if nargin >= 11 && ~isequal(func2str(shFunction), 'getSH'); error('eMagLS:arg', 'only the built-in getSH is accelerated'); end
if nargin < 10 || isempty(shDefinition); shDefinition = 'real'; end
[wMlsL, wMlsR] = emagls_mex('emainch', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
    double(micGridAziRad(:)), order, fs, len, shDefinition);
end
