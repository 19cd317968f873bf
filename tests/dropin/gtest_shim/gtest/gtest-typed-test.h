// part of the GoogleTest stand-in (gtest.h): the typed-test macros live in gtest.h itself
#pragma once
#include "gtest.h"
