// Drop-in for the reference's tools/audioio.hpp: same function names and meaning, implemented by libworldclass_hip.so.
#ifndef WORLD_AUDIOIO_HPP
#define WORLD_AUDIOIO_HPP
#include "world_class_io.h"
#endif
