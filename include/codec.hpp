// Drop-in for the reference's include/codec.hpp: same function names and meaning, computed on the MI355X by
// libworldclass_hip.so (see world_class_codec.h for the device-resident variants).
#ifndef WORLD_CODEC_HPP
#define WORLD_CODEC_HPP
#include "world_class_codec.h"
#endif
