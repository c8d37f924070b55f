// stand-in, see standin.hpp
#pragma once
#include "../standin.hpp"
