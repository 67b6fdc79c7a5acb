/* TEST INFRASTRUCTURE ONLY: empty stand-in, the reference only uses thrust::fill in its launch glue */
#pragma once
