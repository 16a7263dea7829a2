function Y = getCH(N, aziRad, basisType)
% dependencies/getCH.m:1 on the MI355X library
if nargin < 3 || isempty(basisType); basisType = 'real'; end
Y = emagls_mex('ch', N, double(aziRad(:)), basisType);
end
