// taps.h -- host-side filter design (see taps.cpp)
#pragma once
#include <vector>

namespace xrit {
std::vector<float> design_lowpass(double gain, double fs, double cutoff, double transition_width);
std::vector<float> design_rrc(double gain, double fs, double symbol_rate, double alpha, int ntaps);
void design_mmse_table(float *table /* [129*8] */);
}  // namespace xrit
