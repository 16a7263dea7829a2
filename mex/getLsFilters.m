function [wLsL, wLsR] = getLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, shDefinition, shFunction)
% Drop-in replacement of the reference function of the same name, running on the MI355X library.
% A custom shFunction handle cannot cross the C ABI: it is evaluated here and rejected unless it is getSH.
if nargin >= 7 && ~isequal(func2str(shFunction), 'getSH'); error('eMagLS:arg', 'only the built-in getSH is accelerated'); end
if nargin < 6 || isempty(shDefinition); shDefinition = 'real'; end
[wLsL, wLsR] = emagls_mex('ls', double(hL), double(hR), double(hrirGridAziRad(:)), double(hrirGridZenRad(:)), order, shDefinition);
end
