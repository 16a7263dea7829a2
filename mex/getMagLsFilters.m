function [wMlsL, wMlsR] = getMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, fs, len, shDefinition, shFunction)
if nargin >= 9 && ~isequal(func2str(shFunction), 'getSH'); error('eMagLS:arg', 'only the built-in getSH is accelerated'); end
if nargin < 8 || isempty(shDefinition); shDefinition = 'real'; end
[wMlsL, wMlsR] = emagls_mex('magls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), order, fs, len, shDefinition);
end
