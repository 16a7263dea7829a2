function [wMlsL, wMlsR] = getMagLsFilters2D(hLHor, hRHor, horHrirGridAziRad, order, fs, len, chDefinition)
% lib/getMagLsFilters2D.m:1 on the MI355X library
if nargin < 7 || isempty(chDefinition); chDefinition = 'real'; end
[wMlsL, wMlsR] = emagls_mex('magls2d', double(hLHor), double(hRHor), double(horHrirGridAziRad(:)), order, fs, len, chDefinition);
end
