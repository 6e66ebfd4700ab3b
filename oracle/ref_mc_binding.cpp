// Own 3-line binding that exposes the reference's CPU marching cubes
// (/root/reference/src/doubletake/tools/marching_cubes/marching_cubes_cpu.cpp:29) to Python.
// Test infrastructure only; the reference's stock ext.cpp cannot link without its .cu file.
#include <torch/extension.h>
std::tuple<at::Tensor, at::Tensor, at::Tensor> MarchingCubesCpu(const at::Tensor& vol, const float isolevel);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("marching_cubes_cpu", &MarchingCubesCpu); }
