function [wMlsL, wMlsR] = getEMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition, shFunction)
if nargin >= 12 && ~isequal(func2str(shFunction), 'getSH'); error('eMagLS:arg', 'only the built-in getSH is accelerated'); end
if nargin < 11 || isempty(shDefinition); shDefinition = 'real'; end
[wMlsL, wMlsR] = emagls_mex('emagls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
    double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition);
end
