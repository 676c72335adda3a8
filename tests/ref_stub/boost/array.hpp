// Test-side stand-in for <boost/array.hpp> (Boost is not in the image): the reference's ls_extractor/utils.h names
// boost::array<float, 4> in two covariance helpers that the optimiser path never calls.
#pragma once
#include <array>
namespace boost {
template <class T, std::size_t N>
using array = std::array<T, N>;
}
