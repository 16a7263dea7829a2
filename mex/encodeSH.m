function shRecording = encodeSH(smaRecording, micGridAziRad, micGridZenRad, order, shDefinition)
% The two lines verifyEMagLs.m:235-236 on the MI355X library: smaRecording * pinv(getSH(order, micGrid, shDefinition).')
if nargin < 5 || isempty(shDefinition); shDefinition = 'real'; end
shRecording = emagls_mex('encode', double(smaRecording), double(micGridAziRad(:)), double(micGridZenRad(:)), order, shDefinition);
end
