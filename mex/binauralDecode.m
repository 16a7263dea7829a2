function binauralOut = binauralDecode(in, inFs, decodingFilterLeft, decodingFilterRight, decodingFilterFs, compensateDelay, signal, signalFs, horRotAngleRad)
% Core loop of the reference's binauralDecode on the GPU; resampling, rotation and the extra convolution stay in MATLAB.
if decodingFilterFs ~= inFs
    decodingFilterLeft = resample(decodingFilterLeft, inFs, decodingFilterFs);
    decodingFilterRight = resample(decodingFilterRight, inFs, decodingFilterFs);
end
if nargin > 8 && ~isempty(horRotAngleRad) && horRotAngleRad ~= 0; in = rotateHOA_N3D(in, rad2deg(horRotAngleRad), 0, 0); end
comp = nargin > 5 && compensateDelay;
if ~isreal(in) || ~isreal(decodingFilterLeft); error('eMagLS:arg', 'complex-SH rendering is not accelerated yet'); end
binauralOut = emagls_mex('decode', double(in), double(decodingFilterLeft), double(decodingFilterRight), false);
if nargin > 6 && ~isempty(signal)
    if signalFs ~= inFs; signal = resample(signal, inFs, signalFs); end
    binauralOut = [fftfilt(binauralOut(:,1), signal(:,1)), fftfilt(binauralOut(:,2), signal(:,1))];
end
if comp; del = size(decodingFilterLeft,1) / 2; binauralOut = binauralOut(del:end,:); end
end
