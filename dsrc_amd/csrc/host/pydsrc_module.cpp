// pydsrc: the reference's Python module (py/Interface.cpp:55-109, boost::python) over the MI355X host classes, as a
// pybind11 extension.  Same class, method and property names -- FastqRecord, FastqFile, FieldMask, DsrcArchive,
// DsrcModule -- bound in-process to dsrc::wrap::* of dsrc_host.h; every DsrcException surfaces as RuntimeError, which is
// what the reference's exception translator raises (py/Interface.cpp:36-52).
//
// One deliberate difference: the reference binds the SETTER of QualityCompressionLevel to SetDnaCompressionLevel
// (py/Interface.cpp:88,103), so assigning it silently changes the DNA level; here it sets the quality level.
// Extras that the reference does not have: the `Device` property (GPU ordinal) on DsrcArchive / DsrcModule.
#include <pybind11/pybind11.h>

#include "dsrc_host.h"

namespace py = pybind11;
using namespace dsrc::wrap;

namespace
{
template <typename T, typename C>
void BindConfigurable(C& c, bool archive)
{
	c.def_property("LossyCompression", &T::IsLossyCompression, &T::SetLossyCompression)
	 .def_property("DNACompressionLevel", &T::GetDnaCompressionLevel, &T::SetDnaCompressionLevel)
	 .def_property("QualityCompressionLevel", &T::GetQualityCompressionLevel, &T::SetQualityCompressionLevel)
	 .def_property("TagFieldFilterMask", &T::GetTagFieldFilterMask, &T::SetTagFieldFilterMask)
	 .def_property("FastqBufferSizeMB", &T::GetFastqBufferSizeMB, &T::SetFastqBufferSizeMB)
	 .def_property("Crc32Checking", &T::IsCrc32Checking, &T::SetCrc32Checking)
	 .def_property("Device", &T::GetDevice, &T::SetDevice);
	if (archive)
		c.def_property("PlusRepetition", &T::IsPlusRepetition, &T::SetPlusRepetition)
		 .def_property("QualityOffset", &T::GetQualityOffset, &T::SetQualityOffset)
		 .def_property("ColorSpace", &T::IsColorSpace, &T::SetColorSpace);
	else
		c.def_property("ThreadsNumber", &T::GetThreadsNumber, &T::SetThreadsNumber)
		 .def_property("QualityOffset", &T::GetQualityOffset, &T::SetQualityOffset);
}
} // namespace

PYBIND11_MODULE(_pydsrc, m)
{
	m.doc() = "pydsrc on the MI355X path: the reference's Python names over the GPU block compressor / decompressor";
	py::register_exception_translator([](std::exception_ptr p) {
		try { if (p) std::rethrow_exception(p); }
		catch (const dsrc::DsrcException& e) { PyErr_SetString(PyExc_RuntimeError, e.what()); }
	});

	py::class_<FastqRecord>(m, "FastqRecord")
		.def(py::init<>())
		.def_readwrite("tag", &FastqRecord::tag)
		.def_readwrite("sequence", &FastqRecord::sequence)
		.def_readwrite("plus", &FastqRecord::plus)
		.def_readwrite("quality", &FastqRecord::quality);

	py::class_<FastqFile>(m, "FastqFile")
		.def(py::init<>())
		.def("Open", &FastqFile::Open)
		.def("Create", &FastqFile::Create)
		.def("Close", &FastqFile::Close)
		.def("ReadNextRecord", &FastqFile::ReadNextRecord)
		.def("WriteNextRecord", &FastqFile::WriteNextRecord);

	py::class_<FieldMask>(m, "FieldMask")
		.def(py::init<>())
		.def("AddField", &FieldMask::AddField)
		.def("GetMask", &FieldMask::GetMask);

	py::class_<DsrcArchive> archive(m, "DsrcArchive");
	archive.def(py::init<>())
		.def("StartCompress", &DsrcArchive::StartCompress)
		.def("WriteNextRecord", &DsrcArchive::WriteNextRecord)
		.def("FinishCompress", &DsrcArchive::FinishCompress, py::call_guard<py::gil_scoped_release>())
		.def("StartDecompress", &DsrcArchive::StartDecompress)
		.def("ReadNextRecord", &DsrcArchive::ReadNextRecord)
		.def("FinishDecompress", &DsrcArchive::FinishDecompress);
	BindConfigurable<DsrcArchive>(archive, true);

	py::class_<DsrcModule> module(m, "DsrcModule");
	module.def(py::init<>())
		.def("Compress", &DsrcModule::Compress, py::call_guard<py::gil_scoped_release>())
		.def("Decompress", &DsrcModule::Decompress, py::call_guard<py::gil_scoped_release>());
	BindConfigurable<DsrcModule>(module, false);
}
