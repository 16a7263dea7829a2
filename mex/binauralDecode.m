function binauralOut = binauralDecode(in, inFs, decodingFilterLeft, decodingFilterRight, decodingFilterFs, compensateDelay, signal, signalFs, horRotAngleRad)
% Core loop of the reference's binauralDecode (dependencies/binauralDecode.m:33-64) on the GPU, real or complex SH signals and
% filters; resampling, rotation and the extra convolution stay in MATLAB.
if decodingFilterFs ~= inFs
    decodingFilterLeft = resample(decodingFilterLeft, inFs, decodingFilterFs);
    decodingFilterRight = resample(decodingFilterRight, inFs, decodingFilterFs);
end
if nargin > 8 && ~isempty(horRotAngleRad) && horRotAngleRad ~= 0; in = rotateHOA_N3D(in, rad2deg(horRotAngleRad), 0, 0); end
comp = nargin > 5 && compensateDelay;
if isreal(decodingFilterLeft) ~= isreal(decodingFilterRight)      % both real or both complex at the boundary
    decodingFilterLeft = complex(decodingFilterLeft); decodingFilterRight = complex(decodingFilterRight);
end
[binauralOut, imagSum] = emagls_mex('decode', double(in), double(decodingFilterLeft), double(decodingFilterRight), false);
if any(imagSum > 0)     % the warning of :59-63
    warning('Complex binaural output signals (sum of imaginary parts: [%f, %f]). Forcing real outputs.', imagSum(1), imagSum(2));
end
if nargin > 6 && ~isempty(signal)
    if signalFs ~= inFs; signal = resample(signal, inFs, signalFs); end
    binauralOut = [fftfilt(binauralOut(:,1), signal(:,1)), fftfilt(binauralOut(:,2), signal(:,1))];
end
if comp; del = size(decodingFilterLeft,1) / 2; binauralOut = binauralOut(del:end,:); end
end
