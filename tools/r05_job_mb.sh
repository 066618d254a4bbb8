L=audio_denoiser_onnx_amd/libade.so; cp $L /tmp/_keep.so
cp tools/ab/libade_F.so $L
ADE_GRAPH=0 ADE_ROT_DEBUG=1 python tools/debug_melband_rotary.py 2>&1 | grep -v amdgpu | head -40
cp /tmp/_keep.so $L
