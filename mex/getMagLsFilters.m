function [wMlsL, wMlsR] = getMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, fs, len, shDefinition, shFunction)
% lib/getMagLsFilters.m:1-2 on the MI355X library (custom shFunction: evaluated here, see getLsFilters.m)
if nargin < 8 || isempty(shDefinition); shDefinition = 'real'; end
if nargin >= 9 && ~isequal(func2str(shFunction), 'getSH')
    Y = shFunction(order, [hrirGridAziRad(:), hrirGridZenRad(:)], shDefinition);
    [wMlsL, wMlsR] = emagls_mex('magls_y', double(hL), double(hR), Y, order, fs, len, shDefinition);
    return;
end
[wMlsL, wMlsR] = emagls_mex('magls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), order, fs, len, shDefinition);
end
