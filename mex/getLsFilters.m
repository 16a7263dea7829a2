function [wLsL, wLsR] = getLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, shDefinition, shFunction)
% Drop-in replacement of the reference function of the same name (lib/getLsFilters.m:1-2), running on the MI355X library.
% A custom shFunction handle cannot cross the C ABI: it is evaluated here and its matrix is handed over.
if nargin < 6 || isempty(shDefinition); shDefinition = 'real'; end
if nargin >= 7 && ~isequal(func2str(shFunction), 'getSH')
    Y = shFunction(order, [hrirGridAziRad(:), hrirGridZenRad(:)], shDefinition);
    [wLsL, wLsR] = emagls_mex('ls_y', double(hL), double(hR), Y, order, shDefinition);
    return;
end
[wLsL, wLsR] = emagls_mex('ls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), order, shDefinition);
end
