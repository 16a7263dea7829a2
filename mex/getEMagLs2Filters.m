function [wMlsL, wMlsR] = getEMagLs2Filters(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition, shFunction)
% lib/getEMagLs2Filters.m:1-2 on the MI355X library.  The simulation order is max(4, ceil(fs*pi*micRadius/343)) whatever `order`
% is (the reference leaves params.order unset, :51-63): a custom shFunction is evaluated at that order.
if nargin < 11 || isempty(shDefinition); shDefinition = 'real'; end
if nargin >= 12 && ~isequal(func2str(shFunction), 'getSH')
    so = emagls_mex('simorder', 'emagls2', order, fs, micRadius);
    Yh = shFunction(so, [hrirGridAziRad(:), hrirGridZenRad(:)], shDefinition);
    Ym = shFunction(so, [micGridAziRad(:), micGridZenRad(:)], shDefinition);
    [wMlsL, wMlsR] = emagls_mex('emagls2_y', double(hL), double(hR), Yh, micRadius, Ym, order, fs, len, shDefinition);
    return;
end
[wMlsL, wMlsR] = emagls_mex('emagls2', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), micRadius, ...
    double(micGridAziRad(:)), double(micGridZenRad(:)), order, fs, len, shDefinition);
end
