// Drop-in for the reference's include/world_constantnumbers.hpp (reference :1-44): the named constants of namespace `world`
// that callers of the reference may use next to the stage classes.  Values only -- they are the algorithm's published
// parameters (WORLD, M. Morise) and have to be these numbers; nothing here is used by the device path, which carries its own
// copies where it needs them (world_class_amd/csrc).
#ifndef WORLD_CONSTANT_NUMBERS_HPP
#define WORLD_CONSTANT_NUMBERS_HPP

namespace world {

// general
constexpr double kPi = 3.1415926535897932384;
constexpr double kLog2 = 0.69314718055994529;                  // ln 2 as the reference writes it (FFT-size formulas depend on these digits)
constexpr double kEps = 0.00000000000000022204460492503131;    // 2^-52: CheapTrick's infinitesimal noise
constexpr double kMySafeGuardMinimum = 0.000000000001;          // 1e-12: D4C / Synthesis safeguards

// F0 range and defaults (Harvest, CheapTrick)
constexpr double kFloorF0 = 71.0;       // the lowest floor that keeps CheapTrick's FFT at 2048 points at 48 kHz
constexpr double kCeilF0 = 800.0;
constexpr double kDefaultF0 = 500.0;    // stands in for unvoiced frames
constexpr double kMaximumValue = 100000.0;

// D4C
constexpr int kHanning = 1;
constexpr int kBlackman = 2;
constexpr double kFrequencyInterval = 3000.0;
constexpr double kUpperLimit = 15000.0;
constexpr double kThreshold = 0.85;
constexpr double kFloorF0D4C = 47.0;

// codec (mel scale of Stevens & Volkmann, 1940)
constexpr double kM0 = 1127.01048;
constexpr double kF0 = 700.0;
constexpr double kFloorFrequency = 40.0;
constexpr double kCeilFrequency = 20000.0;

}  // namespace world

#endif  // WORLD_CONSTANT_NUMBERS_HPP
