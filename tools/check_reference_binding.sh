#!/bin/bash
# Compile the REFERENCE's own solver templates on Storm::DeviceVector (syntax only): the binding of
# INTEGRATION.md section 3a.  Needs what the reference needs -- a C++20 compiler (g++ >= 12.1 / clang >= 16,
# CMakeLists.txt:79-96) and the include directories of fmt and spdlog (vcpkg.json:6-11), which the build image
# of this repository does not have; run it where StormRuler itself builds:
#
#   STORM_REFERENCE=/path/to/StormRuler FMT_INCLUDE=/usr/include SPDLOG_INCLUDE=/usr/include \
#       tools/check_reference_binding.sh
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${STORM_REFERENCE:?path of a StormRuler checkout}"
: "${FMT_INCLUDE:?directory that holds fmt/format.h}"
: "${SPDLOG_INCLUDE:?directory that holds spdlog/spdlog.h}"
CXX=${CXX:-g++}
for hdr in SolverCg SolverCgs; do   # the solver headers that compile as shipped (SURVEY.md headline fact 5)
  cat > /tmp/storm_hip_binding_$hdr.cpp <<CPP
#define STORM_HIP_NO_SOLVERS 1
#include <Storm/Solvers/$hdr.hpp>
#include <storm_hip/Storm.hpp>
static_assert(Storm::legacy_vector_like<Storm::DeviceVector>);
template class Storm::${hdr#Solver}Solver<Storm::DeviceVector>;   // instantiate every member
int main() { return 0; }
CPP
  "$CXX" -std=c++20 -fsyntax-only -DNDEBUG -I"$ROOT/include" -I"$STORM_REFERENCE/source" -I"$FMT_INCLUDE" \
      -I"$SPDLOG_INCLUDE" /tmp/storm_hip_binding_$hdr.cpp
  echo "$hdr.hpp instantiates on Storm::DeviceVector"
done
