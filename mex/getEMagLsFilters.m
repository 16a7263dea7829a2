function [wMlsL, wMlsR] = getEMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition, shFunction)
% lib/getEMagLsFilters.m:1-2 on the MI355X library.  A custom shFunction is evaluated here at the simulation order
% max(order, ceil(fs*pi*micRadius/343)) (dependencies/getSMAIRMatrix.m:95) on both grids and handed over as matrices.
if nargin < 11 || isempty(shDefinition); shDefinition = 'real'; end
if nargin >= 12 && ~isequal(func2str(shFunction), 'getSH')
    so = emagls_mex('simorder', 'emagls', order, fs, micRadius);
    Yh = shFunction(so, [hrirGridAziRad(:), hrirGridZenRad(:)], shDefinition);
    Ym = shFunction(so, [micGridAziRad(:), micGridZenRad(:)], shDefinition);
    [wMlsL, wMlsR] = emagls_mex('emagls_y', double(hL), double(hR), Yh, micRadius, Ym, order, fs, len, shDefinition);
    return;
end
[wMlsL, wMlsR] = emagls_mex('emagls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
    double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition);
end
