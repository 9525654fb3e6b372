// Drop-in for the reference's tools/parameterio.hpp: same function names and meaning, implemented by libworldclass_hip.so.
#ifndef WORLD_PARAMETERIO_HPP
#define WORLD_PARAMETERIO_HPP
#include "world_class_io.h"
#endif
