#pragma once
// oracle/ref_shim: nothing on the compiled path reads or writes PCD files
