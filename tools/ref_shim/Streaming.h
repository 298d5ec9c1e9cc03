// STAND-IN (empty): tools/ref_shim/README.md
#pragma once
